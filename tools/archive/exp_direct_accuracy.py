"""conv forward error against fp64 for the direct (family 3) and the tiled (family 0) kernel on the same input, at the CWM
level-4-resolution shape of cfg 5 (20x24x28, 24 -> 48 and 48 -> 48): run twice, MODET_CONV_DIRECT=1 / 0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops
torch.manual_seed(0)
for cin, cout in ((24, 48), (48, 48), (128, 128)):
    shape = (20, 24, 28) if cin < 128 else (10, 12, 14)
    x = torch.randn((1,) + shape + (cin,))
    # a realistic CWM input is smooth and far from zero mean: add a large constant part, as InstanceNorm outputs are not
    x = x * 0.05 + torch.linspace(-2, 2, cin)
    w = torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5
    b = torch.randn(cout) * 0.1
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 4, 1)
    y = ops.conv3d_forward(x.cuda().contiguous(), w.cuda(), b.cuda(), False).cpu().double()
    e = (y - ref)
    print(f"DIRECT={os.environ.get('MODET_CONV_DIRECT')} {cin}->{cout}: max|err| {float(e.abs().max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}  mean err {float(e.mean()):+.3e}  max|ref| {float(ref.abs().max()):.2f}")
