"""warp forward / backward at the model's level shapes for the channel counts a project-before-warp reordering would use
(C_in of the level against its projection width): is the scatter's time proportional to its active lanes?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops, synth  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


if __name__ == "__main__":
    out = {}
    for lvl, shape, cs in ((1, (160, 192, 160), (8, 6)), (2, (80, 96, 80), (16, 6)), (3, (40, 48, 40), (32, 12)), (4, (20, 24, 20), (64, 24))):
        amp = 17.0 / 2 ** (lvl - 1) / 4
        fl = torch.from_numpy(synth.make_flow(shape, seed=3, amp=3.0)).cuda().permute(0, 2, 3, 4, 1).contiguous()
        # roughness like the model's flows at random initialisation: add voxel-scale noise
        fl = fl + torch.randn_like(fl) * amp
        for C in cs:
            src = torch.randn(1, *shape, C, device="cuda").requires_grad_(True)
            f = fl.clone().requires_grad_(True)
            o = ops.warp(src, f, 0, False)
            g = torch.randn(1, *shape, C, device="cuda")
            out[f"L{lvl} warp_bwd[C{C}]"] = timeit(lambda: torch.autograd.grad(o, [src, f], g, retain_graph=True))
            with torch.no_grad():
                out[f"L{lvl} warp_fwd[C{C}]"] = timeit(lambda: ops.warp(src, f, 0, False))
            del src, f, o, g
    print(json.dumps(out, indent=1))
