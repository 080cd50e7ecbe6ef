"""Golden vectors for csrc/corr3d.hip from the REAL reference's PR++ Correlation3D
("Baseline methods/PR++/models.py":205-232) -- build container only.

    python tests/golden/make_goldens_corr3d.py

Imports the reference file read-only (its constructor calls .cuda(): mapped to a no-op here), runs it in fp64 on seeded
inputs and writes tests/golden/op_corr3d.npz (inputs, output, gradients); also checks
oracle/modet_torch.py::correlation3d against it and appends the deviation to tests/golden/REPORT.txt."""
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

torch.Tensor.cuda = lambda self, *a, **k: self             # no GPU in the build container
spec = importlib.util.spec_from_file_location("prpp_models", "/root/reference/Baseline methods/PR++/models.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out, lines = {}, []
for tag, C, shape in (("c8", 8, (5, 6, 19)), ("c16", 16, (4, 7, 9)), ("c4", 4, (3, 3, 5))):
    rng = np.random.default_rng(200 + C)
    mov = torch.from_numpy(rng.standard_normal((2, C) + shape).astype(np.float32)).double().requires_grad_(True)
    fix = torch.from_numpy(rng.standard_normal((2, C) + shape).astype(np.float32)).double().requires_grad_(True)
    gy = torch.from_numpy(rng.standard_normal((2, 27) + shape).astype(np.float32)).double()
    m = ref.Correlation3D(C)
    m.w = m.w.double()
    y = m(mov, fix)
    dm, df = torch.autograd.grad(y, [mov, fix], gy)
    err = float((orc.correlation3d(mov, fix) - y).abs().max())
    lines.append(f"{'PR++ Correlation3D C=%d %s' % (C, 'x'.join(map(str, shape))):58s} max|oracle-ref| = {err:.3e}   max|ref| = {float(y.abs().max()):.3e}")
    for n, v in (("mov", mov), ("fix", fix), ("gy", gy), ("out", y), ("dmov", dm), ("dfix", df)):
        out[f"{tag}.{n}"] = v.detach().numpy().astype(np.float64 if n in ("out", "dmov", "dfix") else np.float32)
np.savez_compressed(os.path.join(HERE, "op_corr3d.npz"), **out)
with open(os.path.join(HERE, "REPORT.txt"), "a") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
