#!/usr/bin/env python
"""List the ATen (non-library) GPU ops a train step still issues, with shapes: gradient accumulation adds, copies, fills.
Run on the GPU box:  python tools/prof_aten.py [D,H,W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from smilecode_amd import models, synth
from smilecode_amd.engine import Trainer

shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "160,192,160").split(","))
dev = torch.device("cuda", 0)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
tr = Trainer(model)
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
for _ in range(3):
    tr.train_step(mov, fix, epoch=0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(mov, fix, epoch=0)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = e.cuda_time_total
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.key, e.count, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for dt, k, n, shp in rows[:45]:
    print("%9.1f us  %-28s x%-3d %s" % (dt, k, n, shp))
