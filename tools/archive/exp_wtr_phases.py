"""Phase timing inside conv_wgrad_tr_kernel (tuning build: bash tools/build_variant.sh wtrT conv3d_wtr.hip "-DMODET_TUNING"):
    MODET_HIP_LIB=build/variants/libmodet_hip_wtrT.so python tools/exp_wtr_phases.py Cin Cout level [B]
prints, averaged over waves, the cycles a wave spends per phase of the tile loop."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops, _lib
Cin, Cout, lvl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
D, H, W = (s >> (lvl - 1) for s in (160, 192, 160))
x = torch.randn(B, D, H, W, Cin, device="cuda")
dy = torch.randn(B, D, H, W, Cout, device="cuda")
fn = lambda: ops.conv3d_backward_weight(x, dy, True)
for _ in range(3): fn()
L = _lib.load()
buf = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
L.modet_debug_wtr_timing.argtypes = [ctypes.c_void_p]
assert L.modet_debug_wtr_timing(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
L.modet_debug_wtr_timing(None)
r = buf.view(-1, 8).cpu().double()
r = r[r[:, :7].sum(1) > 0]
t_entry = r[:, 7]
r = r[:, :7]
names = ["global load issue", "barrier 1", "split + LDS write (+ load wait)", "barrier 2", "MFMA phase", "prologue", "epilogue (wave sum, partial store)"]
tot = r.sum(1).mean()
print("wgrad %d->%d L%d B=%d: %d waves, %.0f cycles per wave from entry to exit, call %.1f us; wave entry times span %.0f cycles, last exit - first entry %.0f cycles" % (Cin, Cout, lvl, B, r.shape[0], tot, e0.elapsed_time(e1) * 1e3, t_entry.max() - t_entry.min(), (t_entry + r.sum(1)).max() - t_entry.min()))
for i, n in enumerate(names):
    print("   %-34s %5.1f %%   (%.0f cycles)" % (n, 100 * r[:, i].mean() / tot, r[:, i].mean()))
