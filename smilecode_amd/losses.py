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
    """local (9^3 window) normalized cross correlation loss (reference losses.py:34-94)."""

    def __init__(self, win=None):
        super().__init__()
        if win is not None and list(win) != [9, 9, 9]:
            raise RuntimeError("NCC_vxm: only the default 9x9x9 window is implemented on the HIP path")
        self.win = win

    def forward(self, y_true, y_pred):
        return ops.ncc_loss(y_true.contiguous(), y_pred.contiguous())
