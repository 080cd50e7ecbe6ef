"""Is the two-mode behaviour of back-to-back kernels a property of the box?  A hipGraph of plain library kernels
(fp32 GEMMs + large element-wise passes, no code of this repo), timed like the train step."""
import subprocess
import re
import time

import torch

dev = torch.device("cuda", 0)
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)


def work():
    for _ in range(12):
        c = torch.mm(a, b)
        torch.add(x, 1.0, out=y)
        torch.mul(y, 0.5, out=x)
    return c


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        work()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 150
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
o = subprocess.run(["amd-smi", "metric", "-g", "0", "-p"], capture_output=True, text=True).stdout
for _ in range(100):
    g.replay()
o = subprocess.run(["amd-smi", "metric", "-g", "0", "-p"], capture_output=True, text=True).stdout
torch.cuda.synchronize()
p = re.search(r"SOCKET_POWER: (\d+) W", o)
print("library-kernel graph %.3f ms/replay   power while replaying %s W" % (ms, p.group(1) if p else "?"))
