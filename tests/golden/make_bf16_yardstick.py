"""What does the REFERENCE ITSELF do in bf16?  (build container only: imports /root/reference read-only)

    python tests/golden/make_bf16_yardstick.py [full]

Runs the real reference ModeT (ModeT/models.py) + its losses on CPU twice on the same seeded inputs and weights
(smilecode_amd.synth): plain fp32, and under ``torch.autocast("cpu", dtype=torch.bfloat16)`` -- the way a PyTorch user
would run this network "in bf16" (ATen's autocast policy: conv3d / linear / matmul in bf16, InstanceNorm, LayerNorm,
softmax and grid_sample in fp32; the losses are computed outside the autocast region).  Records how far the autocast run moves away from the fp32 run -- flow rms /
p99.9 / max in voxels, loss, and the relative L2 / cosine of the full parameter gradient -- into
tests/golden/bf16_yardstick.json.  tests/test_gpu_bf16.py states the bf16-STORAGE tolerances of the HIP path (cfg 5) as
multiples of these numbers instead of free-standing constants.  (fp32 itself is within 1e-3 voxels / 4e-15 relative of the
fp64 reference on these inputs -- tests/golden/REPORT.txt -- which is negligible on this scale.)
Fixture = numbers only.  `full` adds BASELINE.json configs[4]'s shape, 160x192x224, sample 0 (~20 GB of host memory)."""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/ModeT")
warnings.filterwarnings("ignore")

import models as ref_models  # noqa: E402  (reference)
import losses as ref_losses  # noqa: E402  (reference)

from smilecode_amd import synth  # noqa: E402

torch.set_num_threads(8)
HEADS = [8, 4, 2, 1, 1]


def ncc_ref(y_true, y_pred):
    """reference NCC_vxm hard-codes .to('cuda') (losses.py:57): patch Tensor.to for the call."""
    orig = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        return orig(self, *a, **k)

    torch.Tensor.to = to
    try:
        return ref_losses.NCC_vxm()(y_true, y_pred)
    finally:
        torch.Tensor.to = orig


def run(shape, autocast):
    m = ref_models.ModeT(shape, head_dim=6, num_heads=HEADS, scale=1)
    sd = m.state_dict()
    for n, v in synth.make_weights(24).items():
        sd[n] = torch.from_numpy(v)
    m.load_state_dict(sd)
    mov, fix = (torch.from_numpy(a) for a in synth.make_pair(shape, 24))
    t0 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        y, flow = m(mov, fix)
    # the losses stay OUT of the autocast region (as any user would do: inside it NCC_vxm's five box-sum convolutions run in
    # bf16 and cross = IJ_sum - u_J I_sum - ... cancels catastrophically: loss error 560 at 64^3)
    loss = ncc_ref(fix, y.float()) + ref_losses.Grad3d(penalty="l2")(flow.float(), fix)
    loss.backward()
    g = {n: p.grad.detach().double().clone() for n, p in m.named_parameters()}
    return flow.detach().double(), float(loss.detach()), g, time.time() - t0


def compare(shape):
    f32, l32, g32, t32 = run(shape, False)
    f16, l16, g16, t16 = run(shape, True)
    ef = f16 - f32
    gv, rv = [], []
    for n, ref in g32.items():
        if float(ref.abs().max()) < 1e-8:              # conv bias under InstanceNorm: analytically zero
            continue
        gv.append(g16[n].reshape(-1)); rv.append(ref.reshape(-1))
    gv, rv = torch.cat(gv), torch.cat(rv)
    return {"shape": list(shape), "flow_absmax": float(f32.abs().max()), "flow_rms": float(ef.pow(2).mean().sqrt()),
            "flow_p999": float(ef.abs().flatten().kthvalue(int(0.999 * ef.numel())).values), "flow_max": float(ef.abs().max()),
            "loss_fp32": l32, "loss_abs_err": abs(l16 - l32), "grad_rel_l2": float((gv - rv).norm() / rv.norm()),
            "grad_cos": float(F.cosine_similarity(gv, rv, 0)), "seconds_fp32": t32, "seconds_autocast": t16}


if __name__ == "__main__":
    shapes = [(32, 48, 32), (64, 64, 64)] + ([(160, 192, 224)] if "full" in sys.argv[1:] else [])
    out_path = os.path.join(HERE, "bf16_yardstick.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    out["what"] = ("reference ModeT (/root/reference/ModeT/models.py + losses.py) under torch.autocast('cpu', bfloat16) vs its own "
                   "fp32 run, synth.make_weights(24) / make_pair(shape, 24); torch %s" % torch.__version__)
    for s in shapes:
        r = compare(s)
        out["x".join(map(str, s))] = r
        print(json.dumps(r), flush=True)
        json.dump(out, open(out_path, "w"), indent=1)
