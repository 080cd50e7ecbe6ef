#!/usr/bin/env python
"""Time every 3x3x3 conv shape of one ModeT train step (160x192x160, B=1) in isolation: forward, dgrad, wgrad.

    python tools/sweep_conv.py [fwd,dgrad,wgrad] [min_level] [iters]
Prints microseconds and useful TFLOP/s per layer (HIP events around `iters` back-to-back launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops

kinds = (sys.argv[1] if len(sys.argv) > 1 else "fwd,dgrad,wgrad").split(",")
min_level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
full = (160, 192, 160)
# (name, level, batch, Cin, Cout)
LAYERS = [("enc0.0", 1, 2, 1, 4), ("enc0.1", 1, 2, 4, 8), ("enc0.2", 1, 2, 8, 8),
          ("enc1.0", 2, 2, 8, 16), ("enc1.1", 2, 2, 16, 16),
          ("enc2.0", 3, 2, 16, 32), ("enc2.1", 3, 2, 32, 32),
          ("enc3.0", 4, 2, 32, 64), ("enc3.1", 4, 2, 64, 64),
          ("enc4.0", 5, 2, 64, 128), ("enc4.1", 5, 2, 128, 128),
          ("cwm5.0", 4, 1, 24, 48), ("cwm5.1", 4, 1, 48, 48), ("cwm5.2", 4, 1, 48, 8),
          ("cwm4.0", 3, 1, 12, 24), ("cwm4.1", 3, 1, 24, 24), ("cwm4.2", 3, 1, 24, 4),
          ("cwm3.0", 2, 1, 6, 12), ("cwm3.1", 2, 1, 12, 12), ("cwm3.2", 2, 1, 12, 2)]
torch.manual_seed(0)
tot = {k: 0.0 for k in kinds}
for name, lvl, B, Cin, Cout in LAYERS:
    if lvl < min_level:
        continue
    D, H, W = (s >> (lvl - 1) for s in full)
    x = torch.randn(B, D, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.1
    b = torch.randn(Cout, device="cuda")
    dy = torch.randn(B, D, H, W, Cout, device="cuda")
    fl = 54.0 * Cin * Cout * B * D * H * W
    line = "%-7s L%d B%d %3d->%-3d %8d vox" % (name, lvl, B, Cin, Cout, B * D * H * W)
    for k in kinds:
        fn = {"fwd": lambda: ops.conv3d_forward(x, w, b, False),
              "dgrad": lambda: ops.conv3d_backward_data(dy, w, Cin),
              "wgrad": lambda: ops.conv3d_backward_weight(x, dy, True)}[k]
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        tot[k] += us
        line += "  %s %7.1f us %5.1f TF" % (k, us, fl / us / 1e6)
    print(line, flush=True)
print("total  " + "  ".join("%s %.1f us" % (k, v) for k, v in tot.items()))
