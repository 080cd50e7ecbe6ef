"""Does the speed of the atomic scatter (level-1 feature-warp backward on the model's own flow) depend on WHERE its buffers live?
Times modet_warp_bwd (d_src + d_flow) with the d_src buffer at many different addresses of one process, and with d_out random vs smooth."""
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
src, fl = rec[8]
B, D, H, W, C = src.shape
st = torch.cuda.current_stream().cuda_stream
dflow = torch.empty_like(fl)
def t(dout, dsrc, n=10):
    def run(): _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), dsrc.data_ptr(), dflow.data_ptr(), B, D, H, W, C, 0, 0, st), "warp_bwd")
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(0)
douts = {"randn": torch.randn(src.shape, device="cuda", generator=g), "zeros": torch.zeros_like(src), "ones": torch.ones_like(src),
         "small": 1e-6 * torch.randn(src.shape, device="cuda", generator=g), "denorm": 1e-41 * torch.ones_like(src)}
keep = []
for name, dout in douts.items():
    ts = []
    for k in range(6):
        keep.append(torch.empty(int((37 + 61 * k) * 2**20 // 4), device="cuda"))      # shift the next allocation
        dsrc = torch.empty_like(src)
        ts.append((dsrc.data_ptr() >> 20, t(dout, dsrc)))
        keep.append(dsrc)
    print("d_out %-7s: " % name + "  ".join("%.3f" % v for _, v in ts) + "   (d_src at MiB " + ",".join(str(a % 100000) for a, _ in ts) + ")", flush=True)
