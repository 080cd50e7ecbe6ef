"""Same-process A/B of two variants of the captured train step (replays on FIXED parameters -- no optimizer step -- so both
variants see identical data; run-to-run noise of separate bench.py processes is ~0.5 %): captures one hipGraph per variant in one process and replays them alternately.

    python tools/ab_graphs.py attr:smilecode_amd.ops.SOME_FLAG=True,False     # a module attribute read at capture time
(used for profiles/r03i_arrival_counter_experiment.txt)
"""
import ast
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd.engine import Trainer  # noqa: E402
from smilecode_amd.models import ModeT  # noqa: E402


def main():
    what = sys.argv[1]
    shape = (160, 192, 160)
    torch.manual_seed(0)
    model = ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((1, 1) + shape, device="cuda", generator=g)
    y = torch.rand((1, 1) + shape, device="cuda", generator=g)
    variants = {}
    if not what.startswith("attr:"):
        raise SystemExit(__doc__)
    target, values = what[5:].split("=")
    modname, attr = target.rsplit(".", 1)
    mod = importlib.import_module(modname)
    for v in values.split(","):
        setattr(mod, attr, ast.literal_eval(v))
        tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
        tr.capture(x, y)
        variants[f"{attr}={v}"] = tr
    for tr in variants.values():
        for _ in range(3):
            tr._graph.replay()
    torch.cuda.synchronize()
    res = {k: [] for k in variants}
    for rep in range(8):
        for name, tr in variants.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                tr._graph.replay()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    for name, v in res.items():
        v.sort()
        print("%-14s median %.3f ms   min %.3f   max %.3f" % (name, v[len(v) // 2], v[0], v[-1]))


if __name__ == "__main__":
    main()
