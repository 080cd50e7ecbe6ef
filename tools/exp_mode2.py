"""Fast (10.6 ms) vs slow (11.9 ms) processes: is it the clock?  Samples rocm-smi while replaying the step, and times a
clock-bound fp32 GEMM and an HBM-bound copy in the same process."""
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
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
            f = re.search(r"fclk clock level: \d+: \((\d+)Mhz\)", out)
            p = re.search(r"Power \(W\): ([\d.]+)", out)
            t = re.findall(r"Temperature \(Sensor (\w+)\) \(C\): ([\d.]+)", out)
            samples.append((int(s.group(1)) if s else -1, int(m.group(1)) if m else -1, int(f.group(1)) if f else -1,
                            float(p.group(1)) if p else -1, t[:3]))
        except Exception as e:  # noqa: BLE001
            samples.append(("err", str(e)))
        time.sleep(0.3)


th = threading.Thread(target=sampler, daemon=True)
th.start()


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("step  %.3f ms" % timed(lambda: tr.train_step(mov, fix), 300), flush=True)
a = torch.randn(8192, 8192, device=dev)
b = torch.randn(8192, 8192, device=dev)
print("sgemm 8192^3 %.3f ms" % timed(lambda: torch.mm(a, b), 20), flush=True)
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
print("copy 1 GiB   %.3f ms" % timed(lambda: dst.copy_(src), 50), flush=True)
print("step  %.3f ms" % timed(lambda: tr.train_step(mov, fix), 100), flush=True)
stop = True
th.join()
print(samples[:3], "...", samples[len(samples) // 2], "...", samples[-2:])
