"""Per-parameter gradient error of the golden train step (tests/golden/e2e_32x48x32.npz) -- which tensors carry
``train.grad_worst_rel_to_max``.  Run once per library build:  MODET_HIP_LIB=build/variants/libmodet_hip_X.so python tools/diag_train_golden.py
Repeats the step a few times so that the float-atomic noise floor is visible beside the build-to-build difference."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth  # noqa: E402
from smilecode_amd.engine import Trainer  # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "e2e_32x48x32.npz"))
shape = (32, 48, 32)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1.0).cuda()
models.load_numpy_weights(m, synth.make_weights(24))
mov, fix = synth.make_pair(shape, 24, 1)
mov, fix = torch.from_numpy(mov).cuda(), torch.from_numpy(fix).cuda()
tr = Trainer(m)
rows = {}
for rep in range(int(os.environ.get("REPS", "3"))):
    tr.fp.zero_grad()
    loss, sim, reg = tr.loss(mov, fix)
    loss.backward()
    for name, p in m.named_parameters():
        ref = g["grad." + name]
        got = p.grad.detach().double().cpu().numpy().reshape(-1)
        got = got if ref.size == got.size else got[::61]
        gmax = float(np.abs(ref).max())
        err = float(np.abs(got - ref.reshape(-1)).max())
        rows.setdefault(name, []).append((err / max(gmax, 1e-30), gmax))
print("lib", os.environ.get("MODET_HIP_LIB", "product"), "loss", float(loss))
order = sorted(rows, key=lambda n: -max(r[0] for r in rows[n] if r[1] > 1e-9) if rows[n][0][1] > 1e-9 else 0)
for n in order[:14]:
    print(f"{n:42s} gmax {rows[n][0][1]:.3e}  rel " + " ".join(f"{r[0]:.2e}" for r in rows[n]))
