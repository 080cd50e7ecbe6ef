"""ctypes binding of libmodet_hip.so (the C ABI declared in include/modet_hip.h).

The product path has NO fallback: if the library is missing or a call fails this module
raises.  ``load()`` is cheap after the first call.  ctypes drops the GIL for the duration
of each call (CDLL), the library itself is stateless/re-entrant.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# MODET_HIP_LIB: load another build of the same library (A/B timing of kernel variants on one box, tools/ab_kernels.py)
LIB_PATH = os.environ.get("MODET_HIP_LIB") or os.path.join(_HERE, "lib", "libmodet_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "modet_hip.h")

P, I, I64, F, SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

class LeafJob(C.Structure):
    """modet_leaf_job_t (include/modet_hip.h)"""
    _fields_ = [("part", P), ("outer", I64), ("outer_stride", I64), ("rows", I64), ("row_stride", I64),
                ("col_group_stride", I64), ("ncols", I), ("col_group", I), ("dst", P * 4), ("n", I * 4)]


# name -> (restype, argtypes); mirrors include/modet_hip.h one to one
SIGNATURES = {
    "modet_hip_version": (I, []),
    "modet_hip_strerror": (C.c_char_p, [I]),
    "modet_qk_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "modet_qk_bwd_ws_bytes": (SZ, [I, I, I, I, I]),
    "modet_qk_bwd": (I, [P, P, P, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_qk_fwd_f64": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "modet_qk_bwd_ws_bytes_f64": (SZ, [I, I, I, I, I]),
    "modet_qk_bwd_f64": (I, [P, P, P, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_na_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, F, P]),
    "modet_na_bwd_ws_bytes": (SZ, [I, I, I, I, I]),
    "modet_na_bwd_partial_rows": (I64, [I, I, I, I, I, I]),
    "modet_na_bwd": (I, [P, P, P, P, P, P, P, P, P, P, SZ, I, I, I, I, I, I, F, P]),
    "modet_na_fwd_t": (I, [P, P, I, P, P, P, I, I, I, I, I, I, F, P]),
    "modet_na_bwd_t": (I, [P, P, I, P, P, P, P, P, P, P, P, SZ, I, I, I, I, I, I, F, P]),
    "modet_conv3d_kernel_family": (I, [I, I, I, I, I, I, I]),
    "modet_conv3d_kernel_family_v": (I, [I, I, I, I, I, I, I, I]),
    "modet_conv3d_ws_bytes": (SZ, [I, I]),
    "modet_conv3d_fwd": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, I, P, P]),
    "modet_conv3d_fwd_bounded": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, I, P, P]),
    "modet_conv3d_stats_bytes": (SZ, [I, I, I, I, I, I]),
    "modet_conv3d_normin_stats_bytes": (SZ, [I, I, I, I, I, I]),
    "modet_conv3d_fwd_stats": (I, [P, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_fwd_stats_bounded": (I, [P, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_fwd_amax_out": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, I, P, P, P]),
    "modet_conv3d_fwd_stats_amax": (I, [P, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P, P]),
    "modet_conv3d_bwd_weight_amax2": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, P, P, P, P]),
    "modet_conv3d_fwd_normin": (I, [P, P, P, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_bwd_data": (I, [P, P, P, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_bwd_weight_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "modet_conv3d_bwd_weight": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_conv3d_bwd_weight_act": (I, [P, P, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_step_ctx_create": (I, [C.POINTER(P)]),
    "modet_step_ctx_destroy": (I, [P]),
    "modet_conv3d_prepack_record": (I, [P, I]),
    "modet_conv3d_prepack_arena_bytes": (SZ, [P]),
    "modet_conv3d_prepack_begin": (I, [P, P, SZ, P]),
    "modet_conv3d_prepack_end": (I, [P]),
    "modet_conv3d_bwd_weight_defer": (I, [P, P, P, P, P, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_wgrad_defer_flush": (I, [P, P]),
    "modet_conv3d_wgrad_defers_operands": (I, [I, I, I, I, I, I]),
    "modet_instnorm_ws_bytes": (SZ, [I, I64, I]),
    "modet_instnorm_lrelu_fwd": (I, [P, P, P, P, P, SZ, I, I64, I, F, P]),
    "modet_instnorm_lrelu_fwd_stats": (I, [P, P, P, P, P, SZ, I, I64, I, F, P]),
    "modet_instnorm_stats": (I, [P, P, P, P, SZ, P, SZ, I, I64, I, F, P]),
    "modet_conv3d_bwd_data_instats_bytes": (SZ, [I, I, I, I, I, I]),
    "modet_conv3d_bwd_data_instats": (I, [P, P, P, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_instnorm_lrelu_bwd_rows": (I, [P, P, P, P, P, P, SZ, P, SZ, I, I64, I, P]),
    "modet_instnorm_lrelu_bwd": (I, [P, P, P, P, P, P, SZ, I, I64, I, P]),
    "modet_lrelu_bwd": (I, [P, P, P, I64, P]),
    "modet_avgpool2_fwd": (I, [P, P, I, I, I, I, I, P]),
    "modet_avgpool2_fwd_x16": (I, [P, P, I, I, I, I, I, P]),
    "modet_instnorm_lrelu_apply_pool": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "modet_instnorm_lrelu_bwd_pool": (I, [P, P, P, I, P, P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_instnorm_lrelu_bwd_amax": (I, [P, P, P, P, P, P, SZ, I, I64, I, P, P]),
    "modet_instnorm_lrelu_bwd_rows_amax": (I, [P, P, P, P, P, P, SZ, P, SZ, I, I64, I, P, P]),
    "modet_instnorm_lrelu_bwd_pool_amax": (I, [P, P, P, I, P, P, P, P, P, SZ, I, I, I, I, I, P, P]),
    "modet_conv3d_bwd_data_amax": (I, [P, P, P, P, SZ, I, I, I, I, I, I, P, P, P]),
    "modet_conv3d_bwd_data_instats_amax": (I, [P, P, P, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P, P]),
    "modet_conv3d_bwd_weight_amax": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, P, P, P]),
    "modet_conv3d_bwd_weight_normin_ok": (I, [I, I, I, I, I, I]),
    "modet_conv3d_bwd_weight_normin": (I, [P, P, P, P, P, P, P, SZ, I, I, I, I, I, I, P, P, P]),
    "modet_avgpool2_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "modet_proj_ln_fwd": (I, [P, P, P, P, P, P, I64, I, I, F, P]),
    "modet_proj_ln_bwd_ws_bytes": (SZ, [I64, I, I]),
    "modet_proj_ln_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, SZ, I64, I, I, F, P]),
    "modet_proj_ln_bwd_pair_ws_bytes": (SZ, [I64, I, I]),
    "modet_proj_ln_fwd_pair": (I, [P, P, P, P, P, P, P, P, I64, I, I, F, P]),
    "modet_proj_ln_bwd_pair_partial_rows": (I64, [I64, I, I]),
    "modet_leaf_reduce_many": (I, [P, I, P]),
    "modet_proj_ln_bwd_pair": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, SZ, I64, I, I, F, P]),
    "modet_proj_ln_fwd_t": (I, [P, I, P, P, P, P, P, I, I64, I, I, F, P]),
    "modet_proj_ln_bwd_pair_t": (I, [P, I, P, P, P, I, P, P, P, P, P, P, P, P, P, P, SZ, I64, I, I, F, P]),
    "modet_warp_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "modet_warp_fwd_o16": (I, [P, P, P, I, I, I, I, I, P]),
    "modet_warp_fwd_t": (I, [P, I, P, P, I, I, I, I, I, I, P]),
    "modet_warp_bwd_t": (I, [P, I, P, P, P, P, I, I, I, I, I, I, I, P]),
    "modet_warp_bwd_acc": (I, [P, I, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "modet_warp_bwd_det_ws_bytes": (SZ, [I, I, I, I, I]),
    "modet_warp_bwd_det": (I, [P, I, P, P, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_warp_bwd_dsrc_tiles_ws_bytes": (SZ, [I, I, I, I, I]),
    "modet_warp_bwd_dsrc_tiles": (I, [P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_warp_bwd_tiles": (I, [P, I, P, P, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_warp_bwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "modet_upsample2_fwd": (I, [P, P, I, I, I, I, I, F, P]),
    "modet_upsample2_bwd": (I, [P, P, I, I, I, I, I, F, P]),
    "modet_upsample2_bwd_sep_ws_bytes": (SZ, [I, I, I, I, I]),
    "modet_upsample2_bwd_sep": (I, [P, P, P, SZ, I, I, I, I, I, F, P]),
    "modet_ncdhw_to_cl": (I, [P, P, I, I, I64, P]),
    "modet_cl_to_ncdhw": (I, [P, P, I, I, I64, P]),
    "modet_cwm_tail_fwd": (I, [P, P, P, I64, I, P]),
    "modet_cwm_tail_bwd": (I, [P, P, P, P, P, I64, I, P]),
    "modet_ncc_ws_bytes": (SZ, [I, I, I, I]),
    "modet_ncc_fwd_bwd": (I, [P, P, P, P, P, SZ, I, I, I, I, P]),
    "modet_ncc_fwd_bwd_win": (I, [P, P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_ncc_box_ws_bytes": (SZ, [I, I, I, I, I, I, I]),
    "modet_ncc_fwd_bwd_box": (I, [P, P, P, P, P, SZ, I, I, I, I, I, I, I, P]),
    "modet_grad3d_ws_bytes": (SZ, [I, I, I, I]),
    "modet_grad3d_fwd_bwd": (I, [P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_grad3d_fwd_bwd_cl": (I, [P, P, P, P, SZ, I, I, I, I, I, F, P]),
    "modet_ncc_fwd_bwd_win_scaled": (I, [P, P, P, P, P, SZ, I, I, I, I, I, F, P]),
    "modet_scale_by_dev_scalar": (I, [P, P, P, I64, P]),
    "modet_adam_amsgrad_step": (I, [P, P, P, P, P, I64, F, F, F, F, I, F, P]),
    "modet_corr3d_ws_bytes": (SZ, [I, I, I, I, I]),
    "modet_corr3d_fwd": (I, [P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_corr3d_bwd": (I, [P, P, P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_label_warp_counts": (I, [P, P, P, P, P, I, I, I, I, P]),
    "modet_jacdet_nonpos_count": (I, [P, P, P, I, I, I, I, P]),
    "modet_conv3d_bf16_ws_bytes": (SZ, [I, I]),
    "modet_conv3d_bf16_stats_bytes": (SZ, [I, I, I, I, I, I]),
    "modet_conv3d_bf16_kernel_family": (I, [I, I, I, I, I, I, I, I]),
    "modet_conv3d_bf16_fwd": (I, [P, I, P, P, P, P, SZ, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_bf16_bwd_data": (I, [P, P, P, I, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_conv3d_bf16_bwd_weight_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "modet_conv3d_bf16_bwd_weight": (I, [P, I, P, P, P, P, SZ, I, I, I, I, I, I, P]),
    "modet_conv3d_bf16_bwd_weight_defer": (I, [P, I, P, P, P, P, SZ, I, I, I, I, I, I, P, P]),
    "modet_instnorm_bf16_ws_bytes": (SZ, [I, I64, I]),
    "modet_instnorm_lrelu_fwd_stats_bf16": (I, [P, P, I, P, P, P, SZ, I, I64, I, F, P]),
    "modet_instnorm_lrelu_bwd_bf16": (I, [P, I, P, P, P, P, P, SZ, I, I64, I, P]),
    "modet_instnorm_lrelu_bwd_pool_bf16": (I, [P, P, P, I, P, P, P, P, P, SZ, I, I, I, I, I, P]),
    "modet_instnorm_lrelu_fwd_stats_pool_bf16": (I, [P, P, P, P, P, P, SZ, I, I, I, I, I, F, P]),
    "modet_cast_bf16": (I, [P, P, I64, I, P]),
}

_lib = None


def header_symbols():
    """Every function name include/modet_hip.h declares (used by the export test)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(modet_[a-z0-9_]+)\s*\(", txt)))


def load():
    """dlopen the library and attach signatures; raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m smilecode_amd.build` "
            "(the ModeT hot path has no CPU / eager fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name) and os.environ.get("MODET_HIP_LIB"):
            continue        # an older build loaded for A/B timing may predate an entry point; the product library may not
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def strerror(code: int) -> str:
    return load().modet_hip_strerror(int(code)).decode()


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed: [{code}] {strerror(code)}")
