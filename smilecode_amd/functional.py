"""``modetqkrpb_cu`` -- the reference's operator boundary (ModeT-cu/functional.py:1-27), on HIP.

Same signature, tensor contract and autograd behaviour as the reference ``ModeTFunction``:
  query (B,heads,D,H,W,d) pre-scaled, key (B,heads,D+2,H+2,W+2,d) zero padded, rpb (heads,3,3,3) or None
  -> attn (B,heads,D,H,W,27); backward returns (d_query, d_key, d_rpb).
``modet_fw`` / ``modet_bw`` mirror the pybind module ``modet`` (ModeT-cu/modet/modet.cpp:4-37).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib
from .ops import _Guard, _p, _stream, _ws


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")            # CHECK_CUDA, utils.h:7
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")               # CHECK_CONTIGUOUS, utils.h:8
    if t.dtype not in (torch.float32, torch.float64):                  # AT_DISPATCH_FLOATING_TYPES, modet_kernel.cu:134
        raise RuntimeError(f'"modet_cu" not implemented for \'{str(t.dtype).replace("torch.", "")}\'')


def _same_dtype(ref, *others):
    for t in others:
        if t is not None and t.dtype != ref.dtype:
            raise RuntimeError(f"expected scalar type {ref.dtype} but found {t.dtype}")   # packed_accessor32<scalar_t>
    return "_f64" if ref.dtype == torch.float64 else ""


def modet_fw(query, key, rpb=None):
    _check_input(query, "query"); _check_input(key, "key")
    B, heads, D, H, W, d = query.shape
    if tuple(key.shape) != (B, heads, D + 2, H + 2, W + 2, d):
        raise RuntimeError("key must be the zero-padded (B,heads,D+2,H+2,W+2,d) tensor")
    if min(D, H, W) < 3:
        raise RuntimeError("Input resolution must be greater than or equal to kernel size.")   # utils.h:10
    if rpb is not None:
        _check_input(rpb, "rpb")
        if rpb.shape[1] != 3:
            raise RuntimeError("modet_fw does not support kernel size %d" % rpb.shape[1])
    sfx = _same_dtype(query, key, rpb)
    attn = torch.empty((B, heads, D, H, W, 27), dtype=query.dtype, device=query.device)
    with _Guard(query):
        fn = getattr(_lib.load(), "modet_qk_fwd" + sfx)
        _lib.check(fn(_p(query), _p(key), _p(rpb), _p(attn), B, heads, D, H, W, d, _stream()), "modet_qk_fwd" + sfx)
    return attn


def modet_bw(d_attn, query, key, biasEnabled):
    _check_input(d_attn, "d_attn"); _check_input(query, "query"); _check_input(key, "key")
    B, heads, D, H, W, d = query.shape
    d_query, d_key = torch.empty_like(query), torch.empty_like(key)
    sfx = _same_dtype(query, key, d_attn)
    d_rpb = torch.empty((heads, 3, 3, 3), dtype=query.dtype, device=query.device) if biasEnabled else None
    L = _lib.load()
    nb = getattr(L, "modet_qk_bwd_ws_bytes" + sfx)(B, heads, D, H, W)
    ws = _ws(nb, query)
    with _Guard(query):
        _lib.check(getattr(L, "modet_qk_bwd" + sfx)(_p(d_attn), _p(query), _p(key), _p(d_query), _p(d_key), _p(d_rpb),
                                                    _p(ws), nb, B, heads, D, H, W, d, _stream()), "modet_qk_bwd" + sfx)
    return [d_query, d_key, d_rpb]


class ModeTFunction(Function):
    @staticmethod
    def forward(ctx, query, key, rpb):
        query = query.contiguous()
        key = key.contiguous()
        attn = modet_fw(query, key, None if rpb is None else rpb.contiguous())
        ctx.save_for_backward(query, key)
        ctx.bias = rpb is not None
        return attn

    @staticmethod
    def backward(ctx, grad_out):
        query, key = ctx.saved_tensors
        d_query, d_key, d_rpb = modet_bw(grad_out.contiguous(), query, key, ctx.bias)
        return d_query, d_key, d_rpb


def modetqkrpb_cu(query, key, rpb):
    return ModeTFunction.apply(query, key, rpb)
