"""The same workload as exp_modes.py through the captured hipGraph (what bench.py times): loss and step time every few steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth
from smilecode_amd.engine import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shape = (160, 192, 160)
dev = torch.device("cuda")
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
for _ in range(5):
    tr.train_step(mov, fix, epoch=0)
tr.capture(mov, fix)
out = []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        loss, sim, reg = tr.train_step(mov, fix, epoch=0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    out.append("%d:%.4f/%.2fms" % (5 + 5 * (i + 1), float(loss), dt))
flat = tr.fp.flat
print(" ".join(out), "| params finite:", bool(torch.isfinite(flat).all()), "grad finite:", bool(torch.isfinite(tr.fp.grad).all()),
      "|grad| max %.3e" % float(tr.fp.grad.abs().max()))
