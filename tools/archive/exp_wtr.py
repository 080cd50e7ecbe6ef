"""conv3d_wtr.hip (bf16x3 weight gradient through LDS transpose reads) at every weight-gradient shape of one train step:
accuracy against ATen-CPU fp64 (small shapes, `check`) and median HIP-event time per call (`time`), to be run once with
MODET_CONV_WTR=0 (exact-f32 kernels) and once with the default.

    python tools/exp_wtr.py check
    MODET_CONV_WTR=0 python tools/exp_wtr.py time out_old.json ; python tools/exp_wtr.py time out_new.json
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops  # noqa: E402

L2, L3, L4, L5 = (80, 96, 80), (40, 48, 40), (20, 24, 20), (10, 12, 10)
# (Cin, Cout, shape, batch): the encoder sees moving + fixed as a batch of 2, the CWM layers one sample
STEP_LAYERS = [(16, 16, L2, 2), (16, 32, L3, 2), (32, 32, L3, 2), (32, 64, L4, 2), (64, 64, L4, 2), (64, 128, L5, 2), (128, 128, L5, 2),
               (6, 12, L2, 1), (12, 12, L2, 1), (12, 2, L2, 1), (12, 24, L3, 1), (24, 24, L3, 1), (24, 4, L3, 1),
               (24, 48, L4, 1), (48, 48, L4, 1), (48, 8, L4, 1), (8, 16, L2, 2)]
if os.environ.get("EXP_WTR_L1"):
    L1 = (160, 192, 160)
    STEP_LAYERS = [(4, 8, L1, 2), (8, 8, L1, 2), (8, 16, L2, 2)]


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


def check():
    worst = 0.0
    cases = [(16, 16, (5, 9, 20), 2), (16, 32, (4, 11, 18), 2), (32, 32, (4, 6, 9), 2), (12, 2, (6, 6, 18), 1), (6, 12, (7, 10, 13), 1),
             (12, 12, (9, 17, 8), 1), (24, 4, (5, 8, 16), 1), (24, 24, (4, 9, 11), 2), (48, 8, (3, 5, 10), 1), (24, 48, (4, 8, 8), 1),
             (64, 128, (2, 3, 10), 2), (128, 128, (3, 4, 5), 2), (4, 8, (6, 9, 17), 2), (8, 8, (7, 8, 9), 1), (20, 20, (5, 6, 7), 1),
             (16, 16, (17, 19, 33), 2)]
    for cin, cout, shape, B in cases:
        gen = torch.Generator().manual_seed(cin * 31 + cout)
        x = torch.randn((B, cin) + shape, generator=gen).double()
        gy = torch.randn((B, cout) + shape, generator=gen).double()
        rw = torch.nn.grad.conv3d_weight(x, (cout, cin, 3, 3, 3), gy, padding=1)
        rb = gy.sum((0, 2, 3, 4))
        xd = x.permute(0, 2, 3, 4, 1).contiguous().float().cuda()
        gd = gy.permute(0, 2, 3, 4, 1).contiguous().float().cuda()
        fam = ops._L().modet_conv3d_kernel_family(B, *shape, cin, cout, 2)
        dw, db = ops.conv3d_backward_weight(xd, gd, True)
        dw2, db2 = ops.conv3d_backward_weight(xd, gd, True)
        ew = float((dw.double().cpu() - rw).abs().max() / rw.abs().max())
        eb = float((db.double().cpu() - rb).abs().max() / rb.abs().max())
        det = torch.equal(dw, dw2) and torch.equal(db, db2)
        worst = max(worst, ew, eb)
        print(f"{cin:3d}->{cout:3d} {shape} B={B} family {fam}: rel err d_w {ew:.2e} d_b {eb:.2e} deterministic {det}", flush=True)
        assert det and ew < 2e-5 and eb < 2e-5, "weight gradient disagrees with fp64"
    print("worst", worst)


def time_all(out):
    rows = []
    tot = 0.0
    for cin, cout, shape, B in STEP_LAYERS:
        g = torch.Generator(device="cuda").manual_seed(cin * 100 + cout)
        x = torch.randn((B,) + shape + (cin,), device="cuda", generator=g)
        dy = torch.randn((B,) + shape + (cout,), device="cuda", generator=g)
        fam = ops._L().modet_conv3d_kernel_family(B, *shape, cin, cout, 2)
        ms = timed(lambda: ops.conv3d_backward_weight(x, dy, True))
        fl = 54.0 * cin * cout * B * shape[0] * shape[1] * shape[2]
        rows.append({"layer": f"{cin}->{cout}", "shape": list(shape), "B": B, "family": fam, "ms": ms, "tflops": fl / ms / 1e9})
        tot += ms
        print("%-9s %-11s B=%d fam %d  %.4f ms  %6.1f TFLOP/s" % (rows[-1]["layer"], "x".join(map(str, shape)), B, fam, ms, fl / ms / 1e9), flush=True)
    print("sum %.4f ms (includes the two reduction launches of every call)" % tot)
    if out:
        json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "check":
        check()
    else:
        time_all(sys.argv[2] if len(sys.argv) > 2 else None)
