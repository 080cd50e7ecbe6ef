"""Which host operations issue the D2D memcpys / memsets of one eager train step (they show up as __amd_rocclr_copyBuffer /
fillBuffer in the kernel trace)?  Lists every CPU operator whose correlated device activities are not kernels we wrote,
with its chain of parent operators (autograd node names included).

    python tools/memcpy_audit.py [D,H,W] > gpurun_out/memcpy_audit.txt
"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd.engine import Trainer  # noqa: E402
from smilecode_amd.models import ModeT  # noqa: E402


def main():
    shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "160,192,160").split(","))
    torch.manual_seed(0)
    model = ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
    tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((1, 1) + shape, device="cuda", generator=g)
    y = torch.rand((1, 1) + shape, device="cuda", generator=g)
    for _ in range(3):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tr.train_step(x, y)
        torch.cuda.synchronize()
    agg = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        if not ev.kernels:
            continue
        if any(c.kernels for c in ev.cpu_children):
            continue
        for k in ev.kernels:
            names[k.name[:60]] += 1
            low = k.name.lower()
            if "memcpy" in low or "memset" in low or "copybuffer" in low or "fillbuffer" in low or k.name.startswith("void at::"):
                chain, p = [], ev
                while p is not None and len(chain) < 6:
                    chain.append(p.name)
                    p = p.cpu_parent
                agg[(k.name[:40], str(ev.input_shapes)[:60], " < ".join(chain)[:200])] += 1
    for key, n in sorted(agg.items(), key=lambda kv: -kv[1]):
        print("%3d  %-40s %-60s %s" % (n, key[0], key[1], key[2]))
    print("--- device activity names")
    for k, n in names.most_common(200):
        if n and ("mem" in k.lower() or "at::" in k):
            print("%4d  %s" % (n, k))


if __name__ == "__main__":
    main()
