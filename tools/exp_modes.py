"""Do the two 'speed modes' of the train step come from the DATA?  Per step of the bench's own workload (seeded weights and pair,
Adam lr 1e-4): loss, |flow| statistics, eager step time and the time of the level-1 feature-warp backward (atomics).
    python tools/exp_modes.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth  # noqa: E402
from smilecode_amd.engine import Trainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    shape = (160, 192, 160)
    dev = torch.device("cuda")
    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
    models.load_numpy_weights(model, synth.make_weights(24))
    tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
    mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
    for i in range(steps):
        tm = ops.KernelTimer(select={"warp_bwd[C8]"})
        ops.set_kernel_timer(tm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, sim, reg = tr.train_step(mov, fix, epoch=0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        ops.set_kernel_timer(None)
        w = sum(v["ms"] for v in tm.summary().values())
        with torch.no_grad():
            model.eval()
            _, flow = model(mov, fix)
            model.train()
        fl = flow.float()
        dx = (fl[..., 1:] - fl[..., :-1]).abs()
        print("step %2d  loss %.5f  sim %.5f  reg %.6f  |flow| max %.3f mean %.4f  |dflow/dx| mean %.4f max %.2f  nan %d   step %.2f ms  warp_bwd[C8] %.3f ms" % (
            i, float(loss), float(sim), float(reg), float(fl.abs().max()), float(fl.abs().mean()), float(dx.mean()), float(dx.max()),
            int(torch.isnan(fl).sum()), dt, w), flush=True)


if __name__ == "__main__":
    main()
