"""Evaluation helpers with the reference's names (ModeT/utils.py:8-106): ``AverageMeter``,
``register_model`` (label / image warp) and ``dice_val_VOI`` -- the Dice-parity harness of
train.py:143-155 / infer.py:86-92, with the label warp and the per-label counting on the GPU."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import ops
from .models import SpatialTransformer


class AverageMeter(object):
    """Computes and stores the average and current value (reference utils.py:8-27)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0
        self.vals = []
        self.std = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.vals.append(val)
        self.std = np.std(self.vals)


class register_model(nn.Module):
    """``register_model(img_size, mode)([img, flow])`` (reference utils.py:74-83)."""

    def __init__(self, img_size=(64, 256, 256), mode="bilinear"):
        super().__init__()
        self.spatial_trans = SpatialTransformer(img_size, mode)

    def forward(self, x):
        img = x[0].cuda().float()
        flow = x[1].cuda()
        return self.spatial_trans(img, flow)


def dice_from_counts(counts, nlabels=54):
    """mean over labels 1..nlabels of 2*n(A&B)/(n(A)+n(B)+1e-5) (arithmetic of reference utils.py:95-105)."""
    c = counts.detach().cpu().numpy().astype(np.float64)
    pred, true, inter = c[0, 1:nlabels + 1], c[1, 1:nlabels + 1], c[2, 1:nlabels + 1]
    return float(np.mean(2.0 * inter / (pred + true + 1e-5)))


def dice_val_VOI(y_pred, y_true, nlabels=54):
    """Dice over the 54 LPBA VOIs of the first batch element (reference utils.py:86-106).
    Inputs are label tensors (B,1,D,H,W); counting runs on the GPU with a zero flow."""
    p = y_pred[0, 0].to(torch.int16).cuda().contiguous()
    t = y_true[0, 0].to(torch.int16).cuda().contiguous()
    zero = torch.zeros((1,) + tuple(p.shape) + (3,), dtype=torch.float32, device=p.device)
    _, counts = ops.label_warp_counts(p, zero, t, nlabels, want_warped=False)
    return np.float64(dice_from_counts(counts, nlabels))


def warp_labels_and_dice(x_seg, flow, y_seg, nlabels=54):
    """fused evaluation tail: nearest-warp ``x_seg`` by ``flow`` (B=1, NCDHW) and score against ``y_seg``.
    Returns (warped labels (1,1,D,H,W) int16, dice).  Equivalent to reg_model([x_seg.float(), flow]) followed by
    dice_val_VOI (train.py:152-153) without the float round trip and the 54 numpy passes."""
    flow_cl = ops.to_channels_last(flow.float().contiguous())
    warped, counts = ops.label_warp_counts(x_seg[0, 0].to(torch.int16).cuda(), flow_cl[:1].contiguous(),
                                           y_seg[0, 0].to(torch.int16).cuda(), nlabels)
    return warped[None, None], dice_from_counts(counts, nlabels)


def jacobian_nonpositive_fraction(flow):
    """fraction of voxels with det(J) <= 0 per sample, as infer.py:89-90 reports it (np.sum(jac_det <= 0) / n_voxels),
    computed on the GPU from the model's flow (B,3,D,H,W): no D2H of the field, one integer per sample comes back."""
    flow_cl = ops.to_channels_last(flow.float().contiguous())
    counts, _ = ops.jacdet_nonpos_count(flow_cl)
    nvox = float(flow.shape[2] * flow.shape[3] * flow.shape[4])
    return [float(c) / nvox for c in counts.tolist()]


def jacobian_determinant_vxm(disp):
    """Jacobian determinant of a (3,D,H,W) displacement field (reference utils.py:108-150): float64 (D,H,W) numpy array,
    computed by the HIP kernel in the reference's operation order (np.gradient differences of flow + identity grid,
    first-row expansion).  Accepts a numpy array or a tensor, as the reference's callers pass `flow.cpu().numpy()[0]`."""
    t = torch.as_tensor(np.ascontiguousarray(disp) if isinstance(disp, np.ndarray) else disp).float().cuda()
    flow_cl = t.permute(1, 2, 3, 0).contiguous()[None]
    _, det = ops.jacdet_nonpos_count(flow_cl, want_det=True)
    return det[0].cpu().numpy()
