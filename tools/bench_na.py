"""Fused neighbourhood attention forward / backward at the level-1 / level-2 shapes (1 head, head_dim 6), through the C ABI
with pre-allocated buffers (no autograd, no allocator: device time only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib  # noqa: E402


def timed(fn, iters=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters


L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for shape in ((160, 192, 160), (80, 96, 80)):
    D, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn((1,) + shape + (6,), device="cuda", generator=g)
    k = torch.randn((1,) + shape + (6,), device="cuda", generator=g)
    rpb = torch.randn((1, 3, 3, 3), device="cuda", generator=g)
    out = torch.empty((1,) + shape + (3,), device="cuda")
    lse = torch.empty((1,) + shape + (1,), device="cuda")
    gy = torch.randn(out.shape, device="cuda", generator=g)
    dq, dk, dr = torch.empty_like(q), torch.empty_like(k), torch.empty_like(rpb)
    nb = L.modet_na_bwd_ws_bytes(1, D, H, W, 1)
    ws = torch.empty(nb // 4 + 2, device="cuda")
    P = lambda t: t.data_ptr()
    fwd = lambda: _lib.check(L.modet_na_fwd(P(q), P(k), P(rpb), P(out), P(lse), 1, D, H, W, 1, 6, 1.0, st), "fwd")
    bwd = lambda: _lib.check(L.modet_na_bwd(P(q), P(k), P(rpb), P(out), P(lse), P(gy), P(dq), P(dk), P(dr), P(ws), nb, 1, D, H, W, 1, 6, 1.0, st), "bwd")
    tf, tb = timed(fwd), timed(bwd)
    n = D * H * W
    print("%-14s na_fwd %.3f ms (%.0f GB/s alg)   na_bwd %.3f ms (%.0f GB/s alg)" % ("x".join(map(str, shape)), tf, 60.0 * n / tf / 1e6, tb, 124.0 * n / tb / 1e6))
