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
    """local normalized cross correlation loss over win^3 windows (reference losses.py:34-94); ``win`` = None (the
    reference's default [9, 9, 9], what train.py:103 uses) or a cubic window [w, w, w] with w in {3, 5, 7, 9}.  The
    reference pads every axis by floor(win[0] / 2) (losses.py:57), so only cubic odd windows keep the volume's shape there."""

    def __init__(self, win=None):
        super().__init__()
        w = [9, 9, 9] if win is None else [int(v) for v in win]
        if len(w) != 3 or w[0] != w[1] or w[0] != w[2] or w[0] not in (3, 5, 7, 9):
            raise RuntimeError(f"NCC_vxm: cubic windows of 3, 5, 7 or 9 voxels are implemented on the HIP path, got {win}")
        self.win = win
        self._w = w[0]

    def forward(self, y_true, y_pred):
        return ops.ncc_loss(y_true.contiguous(), y_pred.contiguous(), self._w)
