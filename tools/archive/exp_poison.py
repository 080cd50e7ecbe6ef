"""Uninitialised-read hunt: every torch.empty / empty_like on the GPU is filled with NaN (float) before use, then one EAGER
train step: any kernel that reads memory it (or a predecessor) never wrote shows up as NaN in the losses or gradients."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_empty, _empty_like = torch.empty, torch.empty_like
POISON = float(os.environ.get("POISON", "nan"))
def empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point(): t.fill_(POISON)
    return t
def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_cuda and t.is_floating_point(): t.fill_(POISON)
    return t
from smilecode_amd import models, ops, synth
from smilecode_amd.engine import Trainer
shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "64,64,64").split(","))
dev = torch.device("cuda")
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
tr._fwd_bwd(mov, fix); torch.cuda.synchronize()
ref = tr.fp.grad.clone()
torch.empty, torch.empty_like = empty, empty_like
out = tr._fwd_bwd(mov, fix); torch.cuda.synchronize()
torch.empty, torch.empty_like = _empty, _empty_like
g = tr.fp.grad.clone()
names = [n for n, _ in model.named_parameters()]
print("losses:", [float(v) for v in out])
bad = 0
for n, (off, k) in zip(names, tr.fp.offsets):
    a, b = ref[off:off + k], g[off:off + k]
    fin = bool(torch.isfinite(b).all())
    d = float((a - b).abs().max()) if fin else float("inf")
    if not (d <= 1e-4 * float(a.abs().max()) + 1e-9):
        bad += 1
        print("  %-34s %7d  ref max %.3e  poisoned-run max %s  diff %.3e" % (n, k, float(a.abs().max()), ("%.3e" % float(b.abs().max())) if fin else "non-finite", d))
print("parameter tensors affected by poisoned allocations:", bad, "of", len(names))
