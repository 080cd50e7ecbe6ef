"""Fast vs slow processes: per-XCD effective GFX clocks and socket power (amd-smi) sampled while the step replays."""
import os
import re
import subprocess
import sys
import threading
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
tr.capture(mov, fix)
samples, stop = [], False


def sampler():
    while not stop:
        out = subprocess.run(["amd-smi", "metric", "-g", "0", "-c", "-p", "-u"], capture_output=True, text=True, timeout=20).stdout
        gfx = [int(x) for x in re.findall(r"GFX_\d+:\s*\n\s*CLK: (\d+) MHz", out)]
        p = re.search(r"SOCKET_POWER: (\d+) W", out)
        thr = re.search(r"THROTTLE_STATUS: (\S+)", out)
        act = re.search(r"UMC_ACTIVITY: (\d+)", out)
        busy = re.search(r"GFX_BUSY_INST:\s*\n\s*XCP_0: \[([^\]]*)\]", out)
        samples.append((gfx, int(p.group(1)) if p else -1, thr.group(1) if thr else "?", int(act.group(1)) if act else -1,
                        busy.group(1) if busy else "?"))
        time.sleep(0.2)


th = threading.Thread(target=sampler, daemon=True)
th.start()
for _ in range(5):
    tr.train_step(mov, fix)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 400
for _ in range(n):
    tr.train_step(mov, fix)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
stop = True
th.join()
mid = samples[2:-1] or samples
clk = [sum(s[0]) / max(len(s[0]), 1) for s in mid]
print("step %.3f ms  samples %d  mean GFX clk %.0f MHz (min %d max %d)  power %.0f W  throttle %s  UMC activity %s" % (
    ms, len(mid), sum(clk) / len(clk), min(min(s[0]) for s in mid), max(max(s[0]) for s in mid),
    sum(s[1] for s in mid) / len(mid), set(s[2] for s in mid), set(s[3] for s in mid)))
print("  e.g.", mid[len(mid) // 2])
print("  UMC activity samples:", [x[3] for x in mid])
