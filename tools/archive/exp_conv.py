"""micro-driver for profiling the L1 conv kernels in isolation: python tools/exp_conv.py [fwd|dgrad|wgrad] [Cin] [Cout] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
B, D, H, W = 2, 160, 192, 160
torch.manual_seed(0)
x = torch.randn(B, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.1
b = torch.randn(Cout, device="cuda")
dy = torch.randn(B, D, H, W, Cout, device="cuda")


def run():
    if what == "fwd":
        return ops.conv3d_forward(x, w, b, False)
    if what == "dgrad":
        return ops.conv3d_backward_data(dy, w, Cin)
    return ops.conv3d_backward_weight(x, dy, True)


for _ in range(2):
    run()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / iters
fl = 54.0 * Cin * Cout * B * D * H * W
print(f"{what} {Cin}->{Cout}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TFLOP/s useful")
