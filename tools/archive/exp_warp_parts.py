"""warp backward at the level-1 feature-warp shape on the model's own flow: both outputs / d_src only / d_flow only"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, models, ops, synth
L = _lib.load()
shape = (160, 192, 160)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda().eval()
models.load_numpy_weights(m, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
rec = {}
orig = ops.warp
def spy(src, flow, mode=0, add_flow=False, flow_bound=0):
    if src.shape[-1] == 8 and not add_flow:
        rec[8] = (src.detach().clone(), flow.detach().clone())
    return orig(src, flow, mode, add_flow, flow_bound)
ops.warp = spy
with torch.no_grad():
    m(mov, fix)
ops.warp = orig
st = torch.cuda.current_stream().cuda_stream
src, fl = rec[8]
B, D, H, W, C = src.shape
dout = torch.randn_like(src)
dsrc, dflow = torch.empty_like(src), torch.empty_like(fl)
def run(ds, df):
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), ds, df, B, D, H, W, C, 0, 0, st), "warp_bwd")
def t(ds, df, n=20):
    for _ in range(3): run(ds, df)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(ds, df)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
print(f"both {t(dsrc.data_ptr(), dflow.data_ptr()):.3f} ms, d_src only {t(dsrc.data_ptr(), None):.3f} ms, d_flow only {t(None, dflow.data_ptr()):.3f} ms (each incl. the 0.03 ms zero fill of d_src where it is written)")

if hasattr(L, "modet_debug_warp_census") or True:
    import ctypes
    try:
        f = L.modet_debug_warp_census
        f.argtypes = [ctypes.c_void_p]
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
        assert f(cnt.data_ptr()) == 0
        run(dsrc.data_ptr(), None); torch.cuda.synchronize()
        f(None)
        n = B * D * H * W * C
        print(f"census: {int(cnt[0])} atomic instructions, {int(cnt[1])} lane-atomics = {int(cnt[1]) / n:.2f} per (voxel, channel), {int(cnt[1]) / max(1, int(cnt[0])):.1f} lanes per instruction")
        d = fl[:, :, :, 1:] - fl[:, :, :, :-1]
        print(f"|flow|max {float(fl.abs().max()):.2f}, |d flow/dx| mean {float(d.abs().mean()):.4f} max {float(d.abs().max()):.3f}")
    except AttributeError:
        pass
