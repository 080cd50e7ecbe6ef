"""PROTOTYPE measurement: the warp backward's d_src on the MODEL'S OWN level-1 flow (C = 8, 160x192x160) --
the shipped kernel (modet_warp_bwd: float atomics with x / y / z merges) against tools/micro/warp_tile_proto.hip
(destination-tile lists + 64-bit fixed-point LDS window + border gather; no global float atomics, deterministic).

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/warp_tile_proto.hip -o build/micro/libwarp_tile.so
    python tools/exp_warp_tile.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smilecode_amd import _lib, models, ops, synth  # noqa: E402

L = _lib.load()
P = ctypes.CDLL(os.path.join(ROOT, "build", "micro", "libwarp_tile.so"))
P.wt_ws_bytes.restype = ctypes.c_size_t
P.wt_ws_bytes.argtypes = [ctypes.c_int] * 3
P.wt_run.restype = ctypes.c_int
P.wt_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]

shape = (160, 192, 160)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda().eval()
models.load_numpy_weights(m, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
rec = {}
orig = ops.warp_tee


def spy(src, flow):
    if src.shape[-1] == 8:
        rec[8] = (src.detach().clone(), flow.detach().clone())
    return orig(src, flow)


ops.warp_tee = spy
with torch.no_grad():
    m(mov, fix)
ops.warp_tee = orig
src, fl = rec[8]
B, D, H, W, C = src.shape
st = torch.cuda.current_stream().cuda_stream
d = fl[:, :, :, 1:] - fl[:, :, :, :-1]
print(f"C={C} {D}x{H}x{W}: |flow|max {float(fl.abs().max()):.2f}, |d flow/dx| mean {float(d.abs().mean()):.4f} max {float(d.abs().max()):.3f}")
torch.manual_seed(0)
dout = torch.randn_like(src)
ref = torch.empty_like(src)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def shipped():
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), ref.data_ptr(), None, B, D, H, W, C, 0, 0, st), "warp_bwd")


print(f"shipped modet_warp_bwd, d_src only (zero-fill + float atomics): {timed(shipped):.3f} ms")
ws = torch.empty(P.wt_ws_bytes(D, H, W) // 4 + 16, dtype=torch.float32, device="cuda")
out = torch.full_like(src, float("nan"))
amax = float(dout.abs().max())


def proto(phases):
    rc = P.wt_run(fl.data_ptr(), dout.data_ptr(), out.data_ptr(), ws.data_ptr(), D, H, W, C, amax, phases, st)
    assert rc == 0, rc


proto(7)
torch.cuda.synchronize()
nt = (D // 8) * (H // 8) * (W // 8)
cnt = ws[:nt].view(torch.int32)
print(f"tiles {nt}, entries {int(cnt.sum())} of {D * H * W}, per tile mean {float(cnt.float().mean()):.0f} max {int(cnt.max())}")
err = float((out - ref).abs().max())
print(f"prototype vs shipped: max |diff| {err:.3e} (max |d_src| {float(ref.abs().max()):.3f}), NaN left: {int(torch.isnan(out).sum())}")
again = torch.full_like(src, float('nan'))
out2, out = out, again
proto(7)
torch.cuda.synchronize()
print("deterministic (two runs bit-identical):", bool(torch.equal(out, out2)))
tb, ta, tc = timed(lambda: proto(1)), timed(lambda: proto(2)), timed(lambda: proto(4))
print(f"prototype: bin (count + scan + fill) {tb:.3f} ms, accumulate {ta:.3f} ms, border gather {tc:.3f} ms, all {timed(lambda: proto(7)):.3f} ms")
print(f"accumulate variants: without the LDS atomics {timed(lambda: proto(2 | 16)):.3f} ms, zero + flush only {timed(lambda: proto(2 | 32)):.3f} ms")
dflow = torch.empty_like(fl)


def shipped_both():
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), ref.data_ptr(), dflow.data_ptr(), B, D, H, W, C, 0, 0, st), "warp_bwd")


def shipped_dflow():
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), None, dflow.data_ptr(), B, D, H, W, C, 0, 0, st), "warp_bwd")


print(f"shipped, same dense random d_out: d_src + d_flow {timed(shipped_both):.3f} ms, d_flow only {timed(shipped_dflow):.3f} ms")
# the step's d_out is not dense: zero it where the fixed image is background (the shipped kernel skips zero contributions)
mask = (fix[0, 0] > 0).unsqueeze(-1).unsqueeze(0).to(dout.dtype)
dout.mul_(mask)
print(f"d_out zero on the fixed image's background ({100 * float(1 - mask.mean()):.0f} % of the voxels): shipped d_src + d_flow "
      f"{timed(shipped_both):.3f} ms, d_src only {timed(shipped):.3f} ms; prototype {timed(lambda: proto(7)):.3f} ms")
