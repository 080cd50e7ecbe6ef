"""Im2Grid CoTr (one-head neighbourhood attention over C channels) on the generic head-dimension kernels: forward and
backward time per pyramid level of a 160x192x160 pair.  python tools/exp_cotr.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models
cotr = models.CoTr().cuda()
for lvl, C in ((1, 8), (2, 16), (3, 32), (4, 64), (5, 128)):
    D, H, W = (s >> (lvl - 1) for s in (160, 192, 160))
    q = torch.randn(1, D, H, W, C, device="cuda", requires_grad=True)
    k = torch.randn(1, D, H, W, C, device="cuda", requires_grad=True)
    gy = torch.randn(1, 3, D, H, W, device="cuda")
    def fwd():
        return cotr(q, k)
    def both():
        torch.autograd.grad(cotr(q, k), [q, k], gy)
    res = []
    for fn in (fwd, both):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    n = D * H * W
    print("level %d C=%3d %8d voxels: fwd %8.1f us (%6.0f GB/s alg)  fwd+bwd %8.1f us" % (lvl, C, n, res[0], n * (2 * C + 3) * 4 / res[0] / 1e3, res[1]))
