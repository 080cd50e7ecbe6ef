"""Golden vectors for the attention kernels at head dimensions other than 6, from the REAL reference's Im2Grid CoTr
("Baseline methods/Im2Grid/models.py":276-322: one head over all C channels, no bias, no scale) -- build container only.

    python tests/golden/make_goldens_cotr.py

Imports the reference file read-only, runs it in fp64 on seeded inputs and writes tests/golden/op_cotr.npz (inputs,
output, gradients); also checks oracle/modet_torch.py::mode_transformer against it and appends the deviation to
tests/golden/REPORT.txt."""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from oracle import modet_torch as orc  # noqa: E402

spec = importlib.util.spec_from_file_location("im2grid_models", "/root/reference/Baseline methods/Im2Grid/models.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out, lines = {}, []
for tag, C, shape in (("c8", 8, (5, 6, 19)), ("c16", 16, (4, 7, 9)), ("c32", 32, (3, 5, 17)), ("c128", 128, (2, 3, 4))):
    rng = np.random.default_rng(100 + C)
    # inputs are rounded to fp32 first (that is what the fixture stores), then the reference runs on them in fp64
    q = torch.from_numpy((rng.standard_normal((2,) + shape + (C,)) * 0.6).astype(np.float32)).double().requires_grad_(True)
    k = torch.from_numpy((rng.standard_normal((2,) + shape + (C,)) * 0.6).astype(np.float32)).double().requires_grad_(True)
    gy = torch.from_numpy(rng.standard_normal((2, 3) + shape).astype(np.float32)).double()
    y = ref.CoTr().double()(q, k)                                # (B,3,H,W,T)
    dq, dk = torch.autograd.grad(y, [q, k], gy)
    yo = orc.mode_transformer(q, k, torch.zeros(1, 27, dtype=torch.float64), 1, 1.0)
    err = float((yo - y).abs().max())
    lines.append(f"{'Im2Grid CoTr C=%d %s' % (C, 'x'.join(map(str, shape))):58s} max|oracle-ref| = {err:.3e}   max|ref| = {float(y.abs().max()):.3e}")
    for n, v in (("q", q), ("k", k), ("gy", gy), ("out", y), ("dq", dq), ("dk", dk)):
        out[f"{tag}.{n}"] = v.detach().numpy().astype(np.float64 if n in ("out", "dq", "dk") else np.float32)
np.savez_compressed(os.path.join(HERE, "op_cotr.npz"), **out)
with open(os.path.join(HERE, "REPORT.txt"), "a") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
