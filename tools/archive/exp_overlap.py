"""dgrad and wgrad of one layer on one stream vs on two streams (do co-resident different kernels fill each other's
MFMA bubbles?):  python tools/exp_overlap.py Cin Cout [level]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops
Cin, Cout = int(sys.argv[1]), int(sys.argv[2])
lvl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = 2
D, H, W = (s >> (lvl - 1) for s in (160, 192, 160))
torch.manual_seed(0)
x = torch.randn(B, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.1
dy = torch.randn(B, D, H, W, Cout, device="cuda")
s2 = torch.cuda.Stream()

def seq():
    ops.conv3d_backward_data(dy, w, Cin)
    ops.conv3d_backward_weight(x, dy, True)

def par():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        ops.conv3d_backward_weight(x, dy, True)
        e2 = torch.cuda.Event(); e2.record()
    ops.conv3d_backward_data(dy, w, Cin)
    torch.cuda.current_stream().wait_event(e2)

for name, fn in (("sequential", seq), ("two streams", par)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    print("%s PERCU=%s %d->%d L%d: %.1f us" % (name, os.environ.get("MODET_CONV_PERCU", "auto"), Cin, Cout, lvl, (time.perf_counter() - t) / 20 * 1e6))
