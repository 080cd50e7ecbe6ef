"""Captured-graph gradients vs eager gradients of the SAME weights at the bench shape: per-parameter comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth
from smilecode_amd.engine import Trainer
shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "160,192,160").split(","))
dev = torch.device("cuda")
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
tr._fwd_bwd(mov, fix)
torch.cuda.synchronize()
ge = tr.fp.grad.clone()
tr.capture(mov, fix)
names = [n for n, _ in model.named_parameters()]
fill = float(os.environ.get("FILL", "nan"))
for rep in range(4):
    if os.environ.get("NOFILL") != "1":
        tr.fp.grad.fill_(fill)
    tr._graph.replay()
    torch.cuda.synchronize()
    print("   static_out (loss, sim, reg):", [float(v) for v in tr._static_out])
    gg = tr.fp.grad.clone()
    bad = []
    for n, (off, k) in zip(names, tr.fp.offsets):
        a, b = ge[off:off + k], gg[off:off + k]
        d = float((a - b).abs().max()) if bool(torch.isfinite(b).all()) else float("inf")
        if not (d <= 1e-3 * float(a.abs().max()) + 1e-9):
            bad.append("%s(%d) eager max %.3e graph max %.3e diff %.3e" % (n, k, float(a.abs().max()), float(b.abs().max()) if bool(torch.isfinite(b).all()) else float("nan"), d))
    print("replay %d: %d of %d parameter tensors differ; |grad| max eager %.3e graph %.3e" % (rep, len(bad), len(names), float(ge.abs().max()), float(gg[torch.isfinite(gg)].abs().max())))
    for l in bad[:12]: print("   ", l)
