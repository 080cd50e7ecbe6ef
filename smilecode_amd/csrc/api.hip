// version / error strings of libmodet_hip.so
#include "common.h"

extern "C" {

int modet_hip_version(void) { return 440;
  /* 0.4.0: + modet_conv3d_kernel_family_v, conv kernel family 4 (conv_wgrad_tr_kernel); no environment reads in product builds
     0.4.1: + typed (fp32 | bf16) entry points modet_na_{fwd,bwd}_t, modet_proj_ln_*_t, modet_warp_*_t, modet_warp_fwd_o16,
              modet_avgpool2_fwd_x16, modet_instnorm_*_pool_bf16; modet_ncc_*_box (any window); conv kernel family 5 (conv_q_kernel)
     0.4.2: + modet_warp_bwd_acc (a second flow gradient added on the way out), modet_warp_bwd_det[_ws_bytes] (deterministic
              fixed-point scatter), modet_conv3d_wgrad_defers_operands (queued multi-layer weight-gradient launches)
     0.4.3: conv kernel families 2 / 4 / 5 on two f16 pieces (forward launches; backward launches given max |d_y|):
            + modet_instnorm_lrelu_bwd{,_rows,_pool}_amax, modet_conv3d_bwd_data{,_instats}_amax, modet_conv3d_bwd_weight_amax,
              MODET_AMAX_SLOTS / _STRIDE / _FLOATS
     0.4.3.1: + modet_grad3d_fwd_bwd_cl (channels-last flow, weighted gradient), modet_ncc_fwd_bwd_win_scaled (weighted gradient):
              a training step seeds its backward with the two loss gradients as the kernels wrote them
     0.4.3.2: + modet_leaf_reduce_many / modet_leaf_job_t, modet_na_bwd_partial_rows, modet_proj_ln_bwd_pair_partial_rows; d_rpb == NULL
              (modet_na_bwd[_t]) and d_Wt == d_bias == d_gamma == d_beta == NULL (modet_proj_ln_bwd_pair[_t]) leave the partial rows
              in the workspace: every leaf reduction of a backward pass in one launch
     0.4.3.3: + modet_proj_ln_fwd_pair; the grouped paired-projection kernels (levels 3-5) run both uses in one launch (grid.y = 2)
     0.4.3.4: + modet_warp_bwd_dsrc_tiles[_ws_bytes]: the warp backward's d_src without global float atomics (destination tiles,
              64-bit fixed-point LDS window); not routed to by default
     0.4.4.0: the destination-tile warp backward rebuilt (payload lists, zero-d_out entries dropped while binning, d_flow in the fill
              pass, 2^-38 fixed point, NaN propagation, hash table that cannot overflow) and made the default of the feature warps;
              modet_warp_bwd_tiles takes (src, src_bf16) and any C % 8 == 0 */ }

const char* modet_hip_strerror(int code) {
  switch (code) {
    case MODET_OK: return "ok";
    case MODET_ERR_NULL: return "required pointer is NULL";
    case MODET_ERR_DIM: return "invalid or inconsistent dimensions";
    case MODET_ERR_UNSUPPORTED: return "configuration not supported by this build";
    case MODET_ERR_WORKSPACE: return "workspace too small (see *_ws_bytes)";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown modet_hip error";
}

}  // extern "C"
