"""Why does the same train step run at 10.6 or 11.9 ms from one process to the next?  Times hipGraph replays of the
step after (a) a plain capture, (b) empty_cache + recapture, (c) carving every buffer out of ONE big cached segment."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth  # noqa: E402
from smilecode_amd.engine import Trainer  # noqa: E402

shape = (160, 192, 160)
dev = torch.device("cuda", 0)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
tr = Trainer(model)
mov, fix = synth.make_pair(shape, 24, 1)
mov, fix = torch.from_numpy(mov).to(dev), torch.from_numpy(fix).to(dev)


def timed(n=40):
    for _ in range(5):
        tr.train_step(mov, fix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(mov, fix)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def recapture():
    tr.release_graph()
    torch.cuda.empty_cache()
    tr.capture(mov, fix)


tr.capture(mov, fix)
print("capture #1            %.3f ms   reserved %.1f GB" % (timed(), torch.cuda.memory_reserved() / 1e9), flush=True)
print("  again               %.3f ms" % timed(), flush=True)
for i in range(3):
    recapture()
    print("empty_cache+recapture %.3f ms   reserved %.1f GB" % (timed(), torch.cuda.memory_reserved() / 1e9), flush=True)
tr.release_graph()
torch.cuda.empty_cache()
big = torch.empty(int(float(os.environ.get("BIG_GB", "48")) * 1e9), dtype=torch.uint8, device=dev)
del big
tr.capture(mov, fix)
print("one big segment first %.3f ms   reserved %.1f GB" % (timed(), torch.cuda.memory_reserved() / 1e9), flush=True)
print("  again               %.3f ms" % timed(), flush=True)
