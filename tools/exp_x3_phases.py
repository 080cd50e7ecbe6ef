"""Phase timing inside conv_x3_kernel (tuning build of the library: MODET_TUNING=1, see tools/x3_phases.sh):
    python tools/exp_x3_phases.py [fwd|dgrad] Cin Cout [level]
prints, averaged over waves, the share of cycles a wave spends in each phase of the plane loop."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops, _lib
what, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lvl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B = 2
D, H, W = (s >> (lvl - 1) for s in (160, 192, 160))
x = torch.randn(B, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.1
b = torch.randn(Cout, device="cuda")
dy = torch.randn(B, D, H, W, Cout, device="cuda")
fn = {"fwd": lambda: ops.conv3d_forward(x, w, b, False), "dgrad": lambda: ops.conv3d_backward_data(dy, w, Cin),
      "wgrad": lambda: ops.conv3d_backward_weight(x, dy, True)}[what]
for _ in range(3): fn()
L = _lib.load()
buf = torch.zeros(16384 * 4 * 6, dtype=torch.int64, device="cuda")
L.modet_debug_x3_timing.argtypes = [ctypes.c_void_p]
assert L.modet_debug_x3_timing(buf.data_ptr()) == 0
fn(); torch.cuda.synchronize()
L.modet_debug_x3_timing(None)
r = buf.view(-1, 6).cpu().double()
r = r[r.sum(1) > 0]
names = ["MFMA (compute)", "flush (stores)", "split + LDS write", "barrier", "global load issue", "wait for loads"]
tot = r.sum(1).mean()
print("%s %d->%d L%d: %d waves, %.0f cycles per wave in the plane loop" % (what, Cin, Cout, lvl, r.shape[0], tot))
for i, n in enumerate(names):
    print("   %-18s %5.1f %%   (%.0f cycles)" % (n, 100 * r[:, i].mean() / tot, r[:, i].mean()))
