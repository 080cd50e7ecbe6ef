"""Where do the small launches of one train step come from?  Runs ONE eager step of the cfg-2 shape under torch.profiler and
lists, per (operator, input shapes, innermost smilecode_amd source line), how many device launches it made -- the ATen
helpers (copy_, add, fill_, zero_, cat ...) that autograd and the host code issue around the C-ABI kernels.

    python tools/launch_audit.py [D,H,W] > gpurun_out/launch_audit.txt
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        tr.train_step(x, y)
        torch.cuda.synchronize()
    agg = collections.Counter()
    dev = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
            continue
        if any(c.kernels for c in ev.cpu_children):          # count the leaf operator that launched, not its parents
            continue
        where = "?"
        for fr in ev.stack or []:
            if "smilecode_amd" in fr or "bench.py" in fr:
                where = fr.split("smilecode_amd/")[-1]
                break
        key = (ev.name, str(ev.input_shapes)[:70], where[:70])
        agg[key] += len(ev.kernels)
        dev[key] += sum(k.duration for k in ev.kernels)
    total = 0
    for key, n in sorted(agg.items(), key=lambda kv: -dev[kv[0]]):
        print("%3d launches %8.1f us  %-16s %-70s %s" % (n, dev[key], key[0], key[1], key[2]))
        total += n
    print("total ATen launches in one step:", total)


if __name__ == "__main__":
    main()
