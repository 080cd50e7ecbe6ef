"""Train / inference step of the hot path with the reference's hyper-parameters
(ModeT/train.py:42-168): NCC + Grad3d('l2') with weights [1,1], Adam(lr 1e-4, amsgrad=True),
poly learning-rate schedule, batch of volume pairs per rank, RCCL gradient all-reduce when >1 rank."""
from __future__ import annotations

import torch

from . import ops
from .losses import Grad3d, NCC_vxm
from .parallel import FlatParams, broadcast_parameters


def poly_lr(epoch, max_epoch=30, init_lr=1e-4, power=0.9):
    """adjust_learning_rate (reference train.py:166-168)"""
    return round(init_lr * (1 - epoch / max_epoch) ** power, 8)


class Trainer:
    def __init__(self, model, lr=1e-4, max_epoch=30, weights=(1.0, 1.0), betas=(0.9, 0.999), eps=1e-8, group=None):
        self.model = model
        self.lr0, self.max_epoch, self.weights = lr, max_epoch, weights
        self.betas, self.eps, self.group = betas, eps, group
        self.fp = FlatParams(model)
        broadcast_parameters(self.fp, 0, group)
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.vmax = torch.zeros_like(self.fp.flat)
        self.step = 0
        self.sim = NCC_vxm()
        self.reg = Grad3d(penalty="l2")

    def loss(self, moving, fixed):
        y_moved, flow = self.model(moving, fixed)
        sim = self.sim(fixed, y_moved) * self.weights[0]
        reg = self.reg(flow, fixed) * self.weights[1]
        return sim + reg, sim, reg

    def train_step(self, moving, fixed, epoch=0):
        """one iteration of train.py:114-133; returns device scalars (no host sync)"""
        self.model.train()
        self.fp.zero_grad()
        loss, sim, reg = self.loss(moving, fixed)
        loss.backward()
        self.fp.gather_grads()
        scale = self.fp.allreduce_grads(self.group)
        self.step += 1
        ops.adam_amsgrad_step_(self.fp.flat, self.fp.grad, self.m, self.v, self.vmax,
                               poly_lr(epoch, self.max_epoch, self.lr0), self.step, self.betas[0], self.betas[1],
                               self.eps, scale)
        return loss.detach(), sim.detach(), reg.detach()

    @torch.no_grad()
    def infer(self, moving, fixed):
        self.model.eval()
        return self.model(moving, fixed)
