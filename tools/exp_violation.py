"""amd-smi violation / throttle accumulators before and after 400 replays of the train step (which mode was it?)."""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth  # noqa: E402
from smilecode_amd.engine import Trainer  # noqa: E402


def viol():
    for args in (["amd-smi", "metric", "-g", "0", "--throttle"], ["amd-smi", "metric", "-g", "0", "-v"]):
        r = subprocess.run(args, capture_output=True, text=True)
        if r.returncode == 0 and len(r.stdout) > 50:
            return r.stdout
    return r.stdout + r.stderr


shape = (160, 192, 160)
dev = torch.device("cuda", 0)
hold = None
if float(os.environ.get("HOLD_GB", "0")) > 0:      # experiment: push every later allocation to another part of the HBM
    hold = torch.empty(int(float(os.environ["HOLD_GB"]) * (1 << 30)), dtype=torch.uint8, device=dev)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
tr = Trainer(model)
mov, fix = synth.make_pair(shape, 24, 1)
mov, fix = torch.from_numpy(mov).to(dev), torch.from_numpy(fix).to(dev)
tr.capture(mov, fix)
for _ in range(10):
    tr.train_step(mov, fix)
torch.cuda.synchronize()
a = viol()
t0 = time.perf_counter()
for _ in range(400):
    tr.train_step(mov, fix)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 400 * 1e3
b = viol()
print("step %.3f ms  (hold %s GB)" % (ms, os.environ.get("HOLD_GB", "0")))
la, lb = a.splitlines(), b.splitlines()
for x, y in zip(la, lb):
    if x != y:
        print("  before:", x.strip(), "| after:", y.strip())
if len(la) < 5:
    print(a[:600])
