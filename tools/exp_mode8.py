"""Is the fast/slow hipGraph-replay mode a function of WHERE the graph's private pool lands in the address space?
Re-captures after shifting the address space with dummy allocations and prints the pool's segment addresses."""
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


def segs():
    ss = [s for s in torch.cuda.memory_snapshot() if s["total_size"] >= (64 << 20)]
    ss.sort(key=lambda s: -s["total_size"])
    return ["%x(%dMB,%s)" % (s["address"], s["total_size"] >> 20, "pool" if s.get("segment_pool_id", (0, 0)) != (0, 0) else "main") for s in ss[:6]]


keep = []
for trial in range(8):
    tr.release_graph()
    torch.cuda.empty_cache()
    if trial:
        keep.append(torch.empty((trial * 37 + 5) << 20, dtype=torch.uint8, device=dev))   # shift what comes next
    tr.capture(mov, fix)
    print("trial %d  %.3f ms  segments %s" % (trial, timed(), segs()), flush=True)
