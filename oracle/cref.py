"""TEST INFRASTRUCTURE ONLY -- builds and binds oracle/modet_ref.c (plain C, fp64).  See that file's header."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "modet_ref.c")
OUT = os.path.join(HERE, "_build", "libmodet_ref.so")
_lib = None
DP, I = C.POINTER(C.c_double), C.c_int


def build(force=False):
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", OUT, SRC, "-lm"], check=True)
    return OUT


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ref_ncc.restype = C.c_double
        _lib.ref_grad3d.restype = C.c_double
        _lib.ref_dice.restype = C.c_double
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(DP)


def modet_fw(q, kpad, rpb):
    B, h, D, H, W, d = q.shape
    q, qp = _d(q); kpad, kp = _d(kpad)
    rp = None
    if rpb is not None:
        rpb, rp = _d(rpb)
    attn = np.empty((B, h, D, H, W, 27))
    lib().ref_modet_fw(qp, kp, rp, attn.ctypes.data_as(DP), B, h, D, H, W, d)
    return attn


def modet_bw(d_attn, q, kpad, bias=True):
    B, h, D, H, W, d = q.shape
    d_attn, ap = _d(d_attn); q, qp = _d(q); kpad, kp = _d(kpad)
    dq, dk = np.empty_like(q), np.empty_like(kpad)
    drpb = np.empty((h, 3, 3, 3)) if bias else None
    lib().ref_modet_bw(ap, qp, kp, dq.ctypes.data_as(DP), dk.ctypes.data_as(DP),
                       drpb.ctypes.data_as(DP) if bias else None, B, h, D, H, W, d)
    return dq, dk, drpb


def na_fwd(q, k, rpb, heads, scale):
    B, D, H, W, Cc = q.shape
    q, qp = _d(q); k, kp = _d(k); rpb, rp = _d(rpb)
    out = np.empty((B, D, H, W, heads * 3))
    lib().ref_na_fwd(qp, kp, rp, out.ctypes.data_as(DP), B, D, H, W, heads, Cc // heads, C.c_double(scale))
    return out


def warp(src, flow, mode=0):
    B, Cc, D, H, W = src.shape
    src, sp = _d(src); flow, fp = _d(flow)
    out = np.empty_like(src)
    lib().ref_warp(sp, fp, out.ctypes.data_as(DP), B, Cc, D, H, W, mode)
    return out


def conv_block(x, w, b, inst_norm):
    B, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    x, xp = _d(x); w, wp = _d(w); b, bp = _d(b)
    raw, out = np.empty((B, Cout, D, H, W)), np.empty((B, Cout, D, H, W))
    lib().ref_conv_block(xp, wp, bp, raw.ctypes.data_as(DP), out.ctypes.data_as(DP), B, Cin, Cout, D, H, W, int(inst_norm))
    return raw, out


def ncc(I_, J_):
    B, _, D, H, W = I_.shape
    I_, ip = _d(I_); J_, jp = _d(J_)
    return float(lib().ref_ncc(ip, jp, B, D, H, W))


def grad3d(flow):
    B, _, D, H, W = flow.shape
    flow, fp = _d(flow)
    return float(lib().ref_grad3d(fp, B, D, H, W))


def dice(pred, truth, nlabels=54):
    p = np.ascontiguousarray(pred, dtype=np.int16)
    t = np.ascontiguousarray(truth, dtype=np.int16)
    return float(lib().ref_dice(p.ctypes.data_as(C.POINTER(C.c_int16)), t.ctypes.data_as(C.POINTER(C.c_int16)),
                                C.c_int64(p.size), nlabels))
