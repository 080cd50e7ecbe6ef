"""What does a train step cost beyond the hipGraph replay?  (the input copies into the graph's static buffers, the Adam launch)
Measured: replay only 10.212 ms, + Adam 10.214, copies + replay 10.220, Trainer.train_step 10.222 -> 0.01 ms: nothing to fold.
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops
from smilecode_amd.engine import Trainer
from smilecode_amd.models import ModeT
shape = (160, 192, 160)
torch.manual_seed(0)
model = ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((1, 1) + shape, device="cuda", generator=g); y = torch.rand((1, 1) + shape, device="cuda", generator=g)
tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1]); tr.capture(x, y)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def adam():
    ops.adam_amsgrad_step_(tr.fp.flat, tr.fp.grad, tr.m, tr.v, tr.vmax, 1e-4, 5, 0.9, 0.999, 1e-8, 1.0)
def copies():
    tr._static_in[0].copy_(x, non_blocking=True); tr._static_in[1].copy_(y, non_blocking=True)
for rep in range(2):
    print("replay only        %.3f" % timeit(lambda: tr._graph.replay()))
    print("replay + adam      %.3f" % timeit(lambda: (tr._graph.replay(), adam())))
    print("copies + replay    %.3f" % timeit(lambda: (copies(), tr._graph.replay())))
    print("train_step         %.3f" % timeit(lambda: tr.train_step(x, y)))
