"""Which LeakyReLU inputs of the golden 32x48x32 model sit on a knife edge?  (test infrastructure: runs the oracle in fp64)

    python tests/golden/kink_report.py

Lists, per encoder block and image, the three smallest |pre-activation| values.  DESIGN.md section 4 (round 5, second session):
the fixed image's `encoder.conv1.2` output has one voxel at 4.1e-7 -- every fp32 evaluation of the model puts it on one side of
LeakyReLU's kink or the other, and the weight gradients of the layers upstream of it differ by ~1e-3 of their maxima between
the two outcomes (tests/test_gpu_e2e.py::test_train_step_golden bounds them at 2e-2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import modet_torch as orc  # noqa: E402
from smilecode_amd import synth  # noqa: E402

shape = (32, 48, 32)
p = {n: torch.from_numpy(v).double() for n, v in synth.make_weights(24).items()}
mov, fix = synth.make_pair(shape, 24, 1)
for tag, img in (("moving", mov), ("fixed", fix)):
    taps = {}
    orc.encoder(p, torch.from_numpy(img).double(), taps, "")
    for k, t in taps.items():
        a = t.flatten()
        pre = torch.where(a >= 0, a, a / 0.1).abs()          # the value LeakyReLU saw
        print(f"{tag:7s} {k:8s} {t.numel():8d} voxel-channels, smallest |pre-activation|:",
              " ".join(f"{float(v):.2e}" for v in torch.sort(pre).values[:3]))
