"""Conv kernels of the full-resolution layers on their own: forward, data gradient, weight gradient at the level-1 / level-2
shapes of one ModeT train step (encoder batch = moving + fixed = 2), median HIP-event time per call.

    MODET_CONV_X3=0 python tools/bench_conv.py      # exact-f32 MFMA kernels
    MODET_CONV_X3=1 python tools/bench_conv.py      # bf16x3 z-marching kernels (default)
    BENCH_CONV_BF16=1 python tools/bench_conv.py    # bf16-storage kernels at the shapes of BASELINE.json configs[4] (160x192x224),
                                                    # with GB/s of the algorithmic bytes (x read once, y written once)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops  # noqa: E402

L1, L2, L3, L4 = (160, 192, 160), (80, 96, 80), (40, 48, 40), (20, 24, 20)
LAYERS = [(4, 8, L1), (8, 8, L1), (8, 16, L2), (16, 16, L2)]
if os.environ.get("BENCH_CONV_LEVELS") == "34":
    LAYERS = [(16, 32, L3), (32, 32, L3), (32, 64, L4), (64, 64, L4)]
if os.environ.get("BENCH_CONV_LEVELS") == "5":      # level 5 of the encoder and the CWM layers at level-4 resolution
    L5 = (10, 12, 10)
    LAYERS = [(64, 128, L5), (128, 128, L5), (24, 48, L4), (48, 48, L4), (48, 8, L4)]


def timed(fn, iters=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main_bf16():
    S1, S2 = (160, 192, 224), (80, 96, 112)
    rows = []
    for cin, cout, shape, inbf in [(4, 8, S1, False), (8, 8, S1, True), (8, 16, S2, True), (16, 16, S2, True)]:
        g = torch.Generator(device="cuda").manual_seed(cin * 100 + cout)
        x = torch.randn((2,) + shape + (cin,), device="cuda", generator=g)
        x = x.bfloat16() if inbf else x
        dy = torch.randn((2,) + shape + (cout,), device="cuda", generator=g).bfloat16()
        w = torch.randn((cout, cin, 3, 3, 3), device="cuda", generator=g) / (27 * cin) ** 0.5
        b = torch.randn((cout,), device="cuda", generator=g)
        n = 2.0 * shape[0] * shape[1] * shape[2]
        isz = 2 if inbf else 4
        r = {"layer": f"{cin}->{cout}", "shape": list(shape), "x_bf16": inbf}
        r["fwd_ms"] = timed(lambda: ops.conv3d_bf16_forward(x, w, b, False))
        r["fwd_stats_ms"] = timed(lambda: ops.conv3d_bf16_forward(x, w, b, True))
        r["dgrad_ms"] = timed(lambda: ops.conv3d_bf16_backward_data(dy, w, cin, inbf))
        r["wgrad_ms"] = timed(lambda: ops.conv3d_bf16_backward_weight(x, dy))
        r["fwd_GBs"] = n * (cin * isz + cout * 2) / r["fwd_ms"] / 1e6
        r["dgrad_GBs"] = n * (cin * isz + cout * 2) / r["dgrad_ms"] / 1e6
        r["wgrad_GBs"] = n * (cin * isz + cout * 2) / r["wgrad_ms"] / 1e6
        rows.append(r)
        print("%-8s %-14s fwd %.3f ms (%4.0f GB/s)  fwd+stats %.3f  dgrad %.3f (%4.0f GB/s)  wgrad %.3f (%4.0f GB/s)" % (
            r["layer"], "x".join(map(str, shape)), r["fwd_ms"], r["fwd_GBs"], r["fwd_stats_ms"], r["dgrad_ms"], r["dgrad_GBs"],
            r["wgrad_ms"], r["wgrad_GBs"]), flush=True)
        del x, dy
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


def main():
    if os.environ.get("BENCH_CONV_BF16") == "1":
        return main_bf16()
    rows = []
    for cin, cout, shape in LAYERS:
        g = torch.Generator(device="cuda").manual_seed(cin * 100 + cout)
        x = torch.randn((2,) + shape + (cin,), device="cuda", generator=g)
        dy = torch.randn((2,) + shape + (cout,), device="cuda", generator=g)
        w = torch.randn((cout, cin, 3, 3, 3), device="cuda", generator=g) / (27 * cin) ** 0.5
        b = torch.randn((cout,), device="cuda", generator=g)
        n = 2.0 * shape[0] * shape[1] * shape[2]
        fl = 54.0 * cin * cout * n
        r = {"layer": f"{cin}->{cout}", "shape": list(shape)}
        r["fwd_ms"] = timed(lambda: ops.conv3d_forward(x, w, b, False))
        r["fwd_stats_ms"] = timed(lambda: ops._Conv3dStats.apply(x, w, b))
        r["dgrad_ms"] = timed(lambda: ops.conv3d_backward_data(dy, w, cin))
        r["wgrad_ms"] = timed(lambda: ops.conv3d_backward_weight(x, dy, True))
        for k in ("fwd", "fwd_stats", "dgrad", "wgrad"):
            r[k + "_TF"] = fl / r[k + "_ms"] / 1e9
        rows.append(r)
        print("%-8s %-14s fwd %.3f ms (%5.1f TF)  fwd+stats %.3f  dgrad %.3f (%5.1f TF)  wgrad %.3f (%5.1f TF)" % (
            r["layer"], "x".join(map(str, shape)), r["fwd_ms"], r["fwd_TF"], r["fwd_stats_ms"], r["dgrad_ms"], r["dgrad_TF"],
            r["wgrad_ms"], r["wgrad_TF"]), flush=True)
        del x, dy
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
