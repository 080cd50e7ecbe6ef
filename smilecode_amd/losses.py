"""``NCC_vxm`` and ``Grad3d`` with the reference's class names and call signatures
(ModeT/losses.py:6-94), computed by the HIP kernels of csrc/losses.hip."""
from __future__ import annotations

import torch

from . import ops


class Grad3d(torch.nn.Module):
    """N-D gradient loss (reference losses.py:6-31): penalty 'l1' (the class default) or 'l2' (what train.py:104 uses)."""

    def __init__(self, penalty="l1", loss_mult=None):
        super().__init__()
        if penalty not in ("l1", "l2"):
            raise RuntimeError(f"Grad3d: unknown penalty {penalty!r} (the reference knows 'l1' and 'l2', losses.py:21)")
        self.penalty = penalty
        self.loss_mult = loss_mult

    def forward(self, y_pred, y_true=None):
        grad = ops.grad3d_loss(y_pred.contiguous(), self.penalty)
        if self.loss_mult is not None:
            grad = grad * self.loss_mult
        return grad


class NCC_vxm(torch.nn.Module):
    """local normalized cross correlation loss over windows of ``win`` voxels (reference losses.py:34-94); ``win`` = None (the
    reference's default [9, 9, 9], what train.py:103 uses) or any [wz, wy, wx].  The reference pads EVERY axis by
    floor(win[0] / 2) (losses.py:57), so an even or anisotropic window averages cc over a grid that differs from the volume's
    by a voxel or more per axis -- reproduced as is.  Cubic windows of 3 / 5 / 7 / 9 voxels run the z-marching kernel, all
    others the general separable path (csrc/losses.hip)."""

    def __init__(self, win=None):
        super().__init__()
        w = [9, 9, 9] if win is None else [int(v) for v in win]
        if len(w) != 3 or min(w) < 1:
            raise RuntimeError(f"NCC_vxm: 3-D volumes take a window of three positive sizes, got {win}")
        self.win = win
        self._w = w

    def forward(self, y_true, y_pred):
        return ops.ncc_loss(y_true.contiguous(), y_pred.contiguous(), self._w)
