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


def jacobian_determinant_vxm(disp):
    """Jacobian determinant of a (3,D,H,W) displacement field with np.gradient, as the reference evaluates it on
    the CPU (utils.py:108-150; `pystrum.volsize2ndgrid` is just the identity index grid).  Post-processing, not part
    of the GPU hot path."""
    disp = np.asarray(disp).transpose(1, 2, 3, 0)
    vol = disp.shape[:-1]
    grid = np.stack(np.meshgrid(*[np.arange(s) for s in vol], indexing="ij"), len(vol))
    J = np.gradient(disp + grid)
    dx, dy, dz = J[0], J[1], J[2]
    d0 = dx[..., 0] * (dy[..., 1] * dz[..., 2] - dy[..., 2] * dz[..., 1])
    d1 = dx[..., 1] * (dy[..., 0] * dz[..., 2] - dy[..., 2] * dz[..., 0])
    d2 = dx[..., 2] * (dy[..., 0] * dz[..., 1] - dy[..., 1] * dz[..., 0])
    return d0 - d1 + d2
