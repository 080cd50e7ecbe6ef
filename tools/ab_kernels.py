"""A/B timing of the level-1/2 hot kernels in isolation (HIP events, median of N launches), for comparing two builds of
libmodet_hip.so on the SAME box (clock / box-to-box variance is ~3 %):

    python tools/ab_kernels.py [--iters 30] [--only conv]             # times smilecode_amd/lib/libmodet_hip.so
    MODET_HIP_LIB=/path/to/other/libmodet_hip.so python tools/ab_kernels.py

Prints one JSON dict {kernel tag: median ms}.  Shapes = the train step's at 160x192x160 (encoder batch of 2)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    out = {"lib": _lib.LIB_PATH}
    L1, L2 = (2, 160, 192, 160), (2, 80, 96, 80)

    def rnd(*s):
        return torch.randn(*s, device=dev)

    def want(tag):
        return not args.only or args.only in tag

    with torch.no_grad():
        for (shape, cin, cout) in ((L1, 4, 8), (L1, 8, 8), (L2, 8, 16), (L2, 16, 16)):
            x, w, b = rnd(*shape, cin).abs_() + 0.3, rnd(cout, cin, 3, 3, 3) * 0.1, rnd(cout)   # |mean| >> std, like level 1
            dy = rnd(*shape, cout)
            lvl = "L1" if shape is L1 else "L2"
            if want("conv"):
                out[f"conv_fwd_stats[{cin}->{cout}]@{lvl}"] = timeit(lambda: ops._Conv3dStats.apply(x, w, b), args.iters)
                out[f"conv_fwd[{cin}->{cout}]@{lvl}"] = timeit(lambda: ops.conv3d_forward(x, w, b, False), args.iters)
                out[f"conv_dgrad[{cout}->{cin}]@{lvl}"] = timeit(lambda: ops.conv3d_backward_data(dy, w, cin), args.iters)
                out[f"conv_wgrad[{cin}->{cout}]@{lvl}"] = timeit(lambda: ops.conv3d_backward_weight(x, dy, True), args.iters)
                if hasattr(ops, "amax_buffer"):              # round 5: the two-f16-piece forms the train step runs (max |d_y| known)
                    am = ops.amax_buffer(dy.abs().max())
                    out[f"conv_dgrad_f16[{cout}->{cin}]@{lvl}"] = timeit(lambda: ops.conv3d_backward_data(dy, w, cin, amax=am), args.iters)
                    out[f"conv_wgrad_f16[{cin}->{cout}]@{lvl}"] = timeit(lambda: ops.conv3d_backward_weight(x, dy, True, amax=am), args.iters)
            del x, dy
        if want("instnorm"):
            x = rnd(*L1, 8)
            out["instnorm_lrelu_fwd[C8]@L1"] = timeit(lambda: ops._InstNormLReLU.apply(x, 1e-5, None), args.iters)
    if want("instnorm"):
        x = rnd(*L1, 8).requires_grad_(True)
        y = ops._InstNormLReLU.apply(x, 1e-5, None)
        g = rnd(*L1, 8)
        out["instnorm_lrelu_bwd[C8]@L1"] = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True), args.iters)
        del x, y, g
    if want("warp"):
        from smilecode_amd import synth
        fl = torch.from_numpy(synth.make_flow((160, 192, 160), seed=3, amp=3.0)).to(dev).permute(0, 2, 3, 4, 1).contiguous()
        for C in (8, 3, 1):
            src = rnd(1, 160, 192, 160, C).requires_grad_(True)
            f = fl.clone().requires_grad_(True)
            o = ops.warp(src, f, 0, False)
            g = rnd(1, 160, 192, 160, C)
            out[f"warp_bwd[C{C}]@L1"] = timeit(lambda: torch.autograd.grad(o, [src, f], g, retain_graph=True), args.iters)
            with torch.no_grad():
                out[f"warp_fwd[C{C}]@L1"] = timeit(lambda: ops.warp(src, f, 0, False), args.iters)
            del src, f, o, g
    if want("na"):
        q, k = rnd(1, 160, 192, 160, 6).requires_grad_(True), rnd(1, 160, 192, 160, 6).requires_grad_(True)
        rpb = (rnd(1, 3, 3, 3) * 0.5).requires_grad_(True)
        o = ops.neighbourhood_attention(q, k, rpb, 1, 1.0)
        g = rnd(1, 160, 192, 160, 3)
        out["na_bwd[h1]@L1"] = timeit(lambda: torch.autograd.grad(o, [q, k, rpb], g, retain_graph=True), args.iters)
        with torch.no_grad():
            out["na_fwd[h1]@L1"] = timeit(lambda: ops.neighbourhood_attention(q, k, rpb, 1, 1.0), args.iters)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
