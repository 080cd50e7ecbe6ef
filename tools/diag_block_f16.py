"""A two-layer ConvInsBlock chain (conv -> InstanceNorm -> LeakyReLU, twice) forward + backward against ATen fp64: which
gradient of the chain a change of the forward convolution's arithmetic moves.  MODET_HIP_LIB selects the build."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops  # noqa: E402


def cl(t):
    return t.float().permute(0, 2, 3, 4, 1).contiguous().cuda()


def ncdhw(t):
    return t.detach().permute(0, 4, 1, 2, 3).double().cpu()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


print("lib", os.environ.get("MODET_HIP_LIB", "product"))
for cin, c, shape, xmode in [(8, 8, (32, 48, 32), "act"), (4, 8, (32, 48, 32), "act"), (8, 16, (16, 24, 16), "act"), (16, 16, (16, 24, 16), "act"),
                             (8, 8, (32, 48, 32), "smooth")]:
    gen = torch.Generator().manual_seed(cin * 7 + c)
    x = torch.randn((2, cin) + shape, generator=gen).double()
    if xmode == "smooth":                                 # spatially smooth input: conv outputs with a small variance around a large mean
        x = F.avg_pool3d(x, 5, 1, 2) * 3 + 1.0
    x = F.leaky_relu(x, 0.1).requires_grad_(True)
    w1 = (torch.randn((c, cin, 3, 3, 3), generator=gen) / np.sqrt(cin * 27)).double().requires_grad_(True)
    w2 = (torch.randn((c, c, 3, 3, 3), generator=gen) / np.sqrt(c * 27)).double().requires_grad_(True)
    b1 = (0.1 * torch.randn(c, generator=gen)).double().requires_grad_(True)
    b2 = (0.1 * torch.randn(c, generator=gen)).double().requires_grad_(True)
    gy = torch.randn((2, c) + shape, generator=gen).double()

    def chain(x, w1, b1, w2, b2):
        h = F.leaky_relu(F.instance_norm(F.conv3d(x, w1, b1, padding=1), eps=1e-5), 0.1)
        return F.leaky_relu(F.instance_norm(F.conv3d(h, w2, b2, padding=1), eps=1e-5), 0.1)

    ref = chain(x, w1, b1, w2, b2)
    r = torch.autograd.grad(ref, [x, w1, w2], gy)
    xd = cl(x.detach()).requires_grad_(True)
    p = [t.detach().float().cuda().requires_grad_(True) for t in (w1, b1, w2, b2)]
    y = ops.instnorm_lrelu(ops.conv3d(ops.instnorm_lrelu(ops.conv3d(xd, p[0], p[1], False)), p[2], p[3], False))
    g = torch.autograd.grad(y, [xd, p[0], p[2]], cl(gy))
    raw = ops.conv3d(xd.detach(), p[0].detach(), p[1].detach(), False)
    rraw = F.conv3d(x, w1, b1, padding=1).detach()
    print(f"{cin}->{c}->{c} {shape} {xmode}: raw1 {rel(ncdhw(raw), rraw):.2e} y {rel(ncdhw(y), ref.detach()):.2e} dx {rel(ncdhw(g[0]), r[0]):.2e} "
          f"dw1 {rel(g[1].double().cpu(), r[1]):.2e} dw2 {rel(g[2].double().cpu(), r[2]):.2e}")
