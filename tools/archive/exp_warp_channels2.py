"""warp backward at level 1 / 2 with the LAST channels of d_out zero (what a zero-padded projected tensor would scatter:
zero contributions are dropped before the atomic): does the scatter's time follow its active lanes?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops, synth  # noqa: E402
from tools.exp_warp_channels import timeit  # noqa: E402,F401

out = {}
for lvl, shape, C, act in ((1, (160, 192, 160), 8, 6), (2, (80, 96, 80), 16, 6), (2, (80, 96, 80), 8, 6)):
    base = torch.from_numpy(synth.make_flow(shape, seed=3, amp=3.0)).cuda().permute(0, 2, 3, 4, 1).contiguous()
    for noise in (0.3, 1.0):
        fl = base + torch.randn_like(base) * noise
        src = torch.randn(1, *shape, C, device="cuda").requires_grad_(True)
        f = fl.clone().requires_grad_(True)
        o = ops.warp(src, f, 0, False)
        g = torch.randn(1, *shape, C, device="cuda")
        g0 = g.clone()
        g0[..., act:] = 0
        out[f"L{lvl} C{C} noise {noise}: full"] = timeit(lambda: torch.autograd.grad(o, [src, f], g, retain_graph=True))
        out[f"L{lvl} C{C} noise {noise}: last {C - act} channels zero"] = timeit(lambda: torch.autograd.grad(o, [src, f], g0, retain_graph=True))
        del src, f, o, g, g0
print(json.dumps(out, indent=1))
