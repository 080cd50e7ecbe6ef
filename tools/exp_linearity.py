"""Batch linearity of the fp32 train-step gradients at cfg 5's shape: grad(batch of 2) vs the mean of the two single-sample
gradients, per parameter (relative to the tensor's max), largest first.  No oracle: ~20 s.
    MODET_CONV_DIRECT=0 python tools/exp_linearity.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth, losses
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "160,192,224").split(","))
w = synth.make_weights(24)
mov_np, fix_np = synth.make_pair(shape, 24, 2)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(m, w)
mov, fix = torch.from_numpy(mov_np).cuda(), torch.from_numpy(fix_np).cuda()
def lg(a, b):
    for p in m.parameters():
        p.grad = None
    y, flow = m(a, b)
    (losses.NCC_vxm()(b, y) + losses.Grad3d(penalty="l2")(flow, b)).backward()
    return {n: p.grad.clone() for n, p in m.named_parameters()}
g0, g1, gb = lg(mov[:1], fix[:1]), lg(mov[1:], fix[1:]), lg(mov, fix)
g0b = lg(mov[:1], fix[:1])
rows = []
for n in gb:
    want = 0.5 * (g0[n] + g1[n])
    gmax = float(want.abs().max())
    if gmax < 1e-8:
        continue
    rows.append((float((gb[n] - want).abs().max()) / gmax, float((g0b[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-30), n, gmax))
rows.sort(reverse=True)
print("env MODET_CONV_DIRECT =", os.environ.get("MODET_CONV_DIRECT"))
for lin, rep, n, gmax in rows[:8]:
    print(f"  lin {lin:.3e}   rerun-of-sample-0 {rep:.3e}   max|g| {gmax:.3e}   {n}")
