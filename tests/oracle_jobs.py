"""The two full-size fp64 runs of the CPU oracle (oracle/modet_torch.py: test infrastructure, never the product path) that
the GPU parity tests compare against, as stand-alone jobs:

    python -m tests.oracle_jobs full160 out.pt      # 160x192x160: fp64 loss + every parameter gradient + flow, fp32 CPU flow
    python -m tests.oracle_jobs cfg5 out.pt         # 160x192x224, 2 samples: sample 0 forward + backward, sample 1 forward

tests/conftest.py starts them as background processes when the collected tests need them, so the ~6 minutes of host CPU
they take (fp64 autograd tapes of 25-40 GB) run BESIDE the GPU tests instead of in front of them (round 3: 645 s of the
driver's 1 200 s limit for `pytest -m gpu`, most of it the GPU idling behind these two runs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HEADS = (8, 4, 2, 1, 1)


def full160():
    from oracle import modet_torch as orc
    from smilecode_amd import synth
    shape = (160, 192, 160)
    w = synth.make_weights(24)
    mov_np, fix_np = synth.make_pair(shape, 24)
    p64 = {n: torch.from_numpy(v).double().requires_grad_(True) for n, v in w.items()}
    loss64, sim64, reg64, _, f64 = orc.train_loss(p64, torch.from_numpy(mov_np).double(), torch.from_numpy(fix_np).double(), HEADS, 6, 1.0)
    g64 = dict(zip(p64, torch.autograd.grad(loss64, list(p64.values()))))
    p32 = {n: torch.from_numpy(v) for n, v in w.items()}
    with torch.no_grad():                                  # the reference's own arithmetic class: ATen-CPU fp32
        _, f32 = orc.modet_forward(p32, torch.from_numpy(mov_np), torch.from_numpy(fix_np), HEADS, 6, 1.0)
    return {"loss": float(loss64), "sim": float(sim64), "reg": float(reg64), "flow64": f64.detach(), "flow32": f32,
            "grad": {n: g.detach() for n, g in g64.items()}}


def cfg5():
    from oracle import modet_torch as orc
    from smilecode_amd import synth
    shape = (160, 192, 224)
    w = synth.make_weights(24)
    mov_np, fix_np = synth.make_pair(shape, 24, 2)
    p64 = {n: torch.from_numpy(v).double().requires_grad_(True) for n, v in w.items()}
    l0, s0, r0, _, f0 = orc.train_loss(p64, torch.from_numpy(mov_np[:1]).double(), torch.from_numpy(fix_np[:1]).double(), HEADS, 6, 1.0)
    g0 = dict(zip(p64, torch.autograd.grad(l0, list(p64.values()))))
    f0 = f0.detach()
    with torch.no_grad():
        _, f1 = orc.modet_forward({n: v.detach() for n, v in p64.items()}, torch.from_numpy(mov_np[1:]).double(),
                                  torch.from_numpy(fix_np[1:]).double(), HEADS, 6, 1.0)
    lab_m = torch.from_numpy(synth.make_labels(shape, 24))[None, None]
    lab_f = torch.from_numpy(synth.make_labels(shape, 25))[None, None]
    dice0 = orc.dice_voi(orc.warp(lab_m.float(), f0.float(), "nearest").long(), lab_f.long())
    return {"flow": torch.cat([f0, f1]), "loss0": float(l0), "sim0": float(s0), "reg0": float(r0),
            "grad0": {n: g.detach() for n, g in g0.items()}, "dice0": dice0}


if __name__ == "__main__":
    what, out = sys.argv[1], sys.argv[2]
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    res = {"full160": full160, "cfg5": cfg5}[what]()
    torch.save(res, out + ".tmp")
    os.replace(out + ".tmp", out)
