"""PR++ Correlation3D kernels: forward / forward+backward time per pyramid level of a 160x192x160 pair."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops
for lvl, C in ((1, 8), (2, 16), (3, 32), (4, 64)):
    D, H, W = (s >> (lvl - 1) for s in (160, 192, 160))
    mov = torch.randn(1, D, H, W, C, device="cuda", requires_grad=True)
    fix = torch.randn(1, D, H, W, C, device="cuda", requires_grad=True)
    gy = torch.randn(1, 27, D, H, W, device="cuda")
    res = []
    for fn in (lambda: ops.correlation3d(mov, fix), lambda: torch.autograd.grad(ops.correlation3d(mov, fix), [mov, fix], gy)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    print("level %d C=%3d %8d voxels: fwd %8.1f us  fwd+bwd %8.1f us" % (lvl, C, D * H * W, res[0], res[1]))
