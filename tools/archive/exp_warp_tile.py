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
P.wt_ws_bytes.argtypes = [ctypes.c_int] * 5
P.wt_run.restype = ctypes.c_int
P.wt_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]

shape = (160, 192, 160)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda().eval()
models.load_numpy_weights(m, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
rec = {}
orig = ops.warp_tee


def spy(src, flow):
    if src.shape[-1] in (8, 16, 32):
        rec[src.shape[-1]] = (src.detach().clone(), flow.detach().clone())
    return orig(src, flow)


ops.warp_tee = spy
with torch.no_grad():
    m(mov, fix)
ops.warp_tee = orig
st = torch.cuda.current_stream().cuda_stream


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


def case(src, fl, label, sparse_mask=None, detail=False):
    B, D, H, W, C = src.shape
    d = fl[:, :, :, 1:] - fl[:, :, :, :-1]
    print(f"--- {label}: B={B} C={C} {D}x{H}x{W}: |flow|max {float(fl.abs().max()):.2f}, |d flow/dx| mean {float(d.abs().mean()):.4f} max {float(d.abs().max()):.3f}")
    torch.manual_seed(0)
    dout = torch.randn_like(src)
    if sparse_mask is not None:
        dout.mul_(sparse_mask)
    ref, dflow = torch.empty_like(src), torch.empty_like(fl)

    def shipped(ds=True, df=False):
        _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), ref.data_ptr() if ds else None,
                                    dflow.data_ptr() if df else None, B, D, H, W, C, 0, 0, st), "warp_bwd")

    ws = torch.empty(P.wt_ws_bytes(B, D, H, W, C) // 4 + 16, dtype=torch.float32, device="cuda")
    out = torch.full_like(src, float("nan"))

    def proto(phases, dst=None):
        rc = P.wt_run(fl.data_ptr(), dout.data_ptr(), (out if dst is None else dst).data_ptr(), ws.data_ptr(), B, D, H, W, C, phases, st)
        assert rc == 0, rc

    shipped()
    proto(7)
    torch.cuda.synchronize()
    err = float((out - ref).abs().max())
    out2 = torch.full_like(src, float("nan"))
    proto(7, out2)
    torch.cuda.synchronize()
    print(f"    prototype vs shipped: max |diff| {err:.3e} of max |d_src| {float(ref.abs().max()):.3f}; NaN left {int(torch.isnan(out).sum())}; "
          f"two runs bit-identical: {bool(torch.equal(out, out2))}")
    print(f"    shipped: d_src {timed(shipped):.3f} ms, d_src + d_flow {timed(lambda: shipped(True, True)):.3f}, d_flow only "
          f"{timed(lambda: shipped(False, True)):.3f} | prototype d_src {timed(lambda: proto(7)):.3f} ms")
    if detail:
        print(f"    prototype by pass: bin {timed(lambda: proto(1)):.3f}, absmax + accumulate {timed(lambda: proto(2)):.3f} "
              f"(no LDS atomics {timed(lambda: proto(2 | 16)):.3f}, zero + flush only {timed(lambda: proto(2 | 32)):.3f}), border {timed(lambda: proto(4)):.3f}")
    # the sequence is kernels with fixed arguments: capture it and replay
    g = torch.cuda.CUDAGraph()
    out.fill_(float("nan"))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sst = side.cuda_stream
        with torch.cuda.graph(g, stream=side):
            rc = P.wt_run(fl.data_ptr(), dout.data_ptr(), out.data_ptr(), ws.data_ptr(), B, D, H, W, C, 7, sst)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        out.fill_(float("nan"))
        g.replay()
    torch.cuda.synchronize()
    print(f"    hipGraph replay == eager result: {bool(torch.equal(out, out2))}")


src8, fl8 = rec[8]
case(src8, fl8, "level 1, dense random d_out", detail=True)
mask = (fix[0, 0] > 0).unsqueeze(-1).unsqueeze(0).to(src8.dtype)
case(src8, fl8, "level 1, d_out zero on the fixed image's background (%d %% of the voxels)" % round(100 * float(1 - mask.mean())), sparse_mask=mask)
if 16 in rec:
    case(*rec[16], "level 2 (two channel passes)")
case(torch.cat([src8[:, :80], src8[:, 80:]], 0).contiguous(), torch.cat([fl8[:, :80], fl8[:, 80:]], 0).contiguous(), "two samples (the level-1 volume cut in two along z)")
