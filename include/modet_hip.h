/* libmodet_hip.so -- C ABI of the MI355X-native ModeT hot path (gfx950, hand-written HIP).
 *
 * Drop-in boundary for ZAX130/SmileCode's ModeT operator layer.  The reference binds ONE
 * native module, `modet` (ModeT-cu/modet/modet.cpp:34-37: modet_fw / modet_bw), from Python
 * (ModeT-cu/functional.py:3,10,18) and gets every other op of ModeT.forward
 * (ModeT/models.py:377-412) from ATen.  This library exports that operator (modet_qk_*) with
 * the reference's exact tensor contract, plus the fused/native forms of every other op on
 * the path; `smilecode_amd/` binds it with ctypes (INTEGRATION.md shows the reference-side stub).
 *
 * Conventions
 *  - plain C, no torch types: raw DEVICE pointers, sizes, an explicit hipStream_t (as void*).
 *  - fp32 unless the name says otherwise (*_bf16: bf16 activations, fp32 accumulate; *_f64: the operator boundary in
 *    double).  Activations are channels-last "(B,D,H,W,C)" unless a comment says
 *    otherwise; (D,H,W) are the reference's (H,W,T) = the three spatial axes in memory order.
 *  - caller allocates every output and workspace; the library never allocates, frees or retains
 *    device memory and is re-entrant (the autograd engine calls the backward entry points from
 *    another thread).  Its only process-wide state is a mutex-guarded memo of per-kernel occupancy
 *    constants (filled on first use, never changed afterwards).  The step-batching entry points (recorded weight-packing
 *    jobs, queued weight-gradient reductions) keep their tables in a CALLER-OWNED context, modet_step_ctx_t: host
 *    memory only, one per trainer / stream of work, passed explicitly; NULL = "no batching" wherever it is optional.
 *  - every entry point only enqueues stream work (kernels, hipMemsetAsync): a sequence of calls
 *    can be captured into a hipGraph once each kernel has been launched at least once.
 *  - all launches are asynchronous on `stream`; nothing synchronises.
 *  - return value: 0 = ok, <0 = argument error (enum below), >0 = hipError_t from the launch.
 *  - `ws`/`ws_bytes`: scratch from the matching *_ws_bytes(); contents undefined afterwards.
 *  - 64-bit safe offsets throughout (the reference's PackedTensorAccessor32 limit,
 *    modet_kernel.cu:19-22, does not apply).
 */
#ifndef MODET_HIP_H
#define MODET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* modet_stream_t; /* hipStream_t */

/* Caller-owned context of the step-batching entry points (modet_conv3d_prepack_*, modet_conv3d_*_defer,
 * modet_conv3d_wgrad_defer_flush).  Host memory only: job descriptions (pointers + geometry), no device memory, no
 * stream.  Calls on ONE context may come from two threads (forward thread, autograd thread; guarded inside); different
 * contexts never interact, so two trainers in one process -- two threads, two devices, a training and an evaluation
 * model -- each use their own. */
typedef struct modet_step_ctx modet_step_ctx_t;
int modet_step_ctx_create(modet_step_ctx_t** out);
int modet_step_ctx_destroy(modet_step_ctx_t* ctx);   /* NULL is a no-op; nothing queued may still be in flight */

enum {
  MODET_OK = 0,
  MODET_ERR_NULL = -1,        /* required pointer is NULL (reference: CHECK_CUDA, utils.h:7)          */
  MODET_ERR_DIM = -2,         /* non-positive / inconsistent dims (reference: CHECK_3DFEATMAP, utils.h:10) */
  MODET_ERR_UNSUPPORTED = -3, /* configuration not built (reference: CHECK_KERNELSIZE, utils.h:11-14)  */
  MODET_ERR_WORKSPACE = -4    /* ws_bytes smaller than *_ws_bytes() says                               */
};

int modet_hip_version(void);
const char* modet_hip_strerror(int code);

/* ---------------------------------------------------------------------------------------------
 * Neighbourhood attention, reference operator contract  (replaces modet_fw / modet_bw,
 * ModeT-cu/modet/modet.cpp:4-31 -> modet_kernel.cu:17-381).
 *   q    (B,heads,D,H,W,hd)        already multiplied by scale (ModeT-cu/models.py:304)
 *   kpad (B,heads,D+2,H+2,W+2,hd)  zero padded by the caller (models.py:309-310)
 *   rpb  (heads,3,3,3) or NULL (= zeros, modet.cpp:13)
 *   attn (B,heads,D,H,W,27)        token t = 9*ki+3*kj+kk (modet_kernel.cu:44-52,:80)
 * Backward: d_q like q, d_kpad like kpad (pad ring included, as the reference returns it),
 * d_rpb (heads,27) or NULL when the bias is disabled (modet_kernel.cu:343-345).
 * d_rpb is reduced in two deterministic stages (the reference uses fastAtomicAdd, :315). */
int modet_qk_fwd(const float* q, const float* kpad, const float* rpb, float* attn,
                 int B, int heads, int D, int H, int W, int hd, modet_stream_t stream);
size_t modet_qk_bwd_ws_bytes(int B, int heads, int D, int H, int W);
int modet_qk_bwd(const float* d_attn, const float* q, const float* kpad,
                 float* d_q, float* d_kpad, float* d_rpb, void* ws, size_t ws_bytes,
                 int B, int heads, int D, int H, int W, int hd, modet_stream_t stream);
/* double instantiation of the same operator: the reference dispatches float and double
 * (AT_DISPATCH_FLOATING_TYPES, modet_kernel.cu:134 and :364); same contract, fp64 arithmetic throughout. */
int modet_qk_fwd_f64(const double* q, const double* kpad, const double* rpb, double* attn,
                     int B, int heads, int D, int H, int W, int hd, modet_stream_t stream);
size_t modet_qk_bwd_ws_bytes_f64(int B, int heads, int D, int H, int W);
int modet_qk_bwd_f64(const double* d_attn, const double* q, const double* kpad,
                     double* d_q, double* d_kpad, double* d_rpb, void* ws, size_t ws_bytes,
                     int B, int heads, int D, int H, int W, int hd, modet_stream_t stream);

/* Fused ModeTransformer.forward (ModeT/models.py:308-334 == ModeT-cu/models.py:300-316):
 *   logits = scale*q.k(n+off) + rpb, softmax over the 27 modes, out = sum_t p[t]*off(t).
 *   q,k (B,D,H,W,heads*hd) channels-last, unpadded, unscaled; rpb (heads,27);
 *   out (B,D,H,W,heads*3), channel = head*3+axis (models.py:332).  hd = 6 (train.py:49) takes the specialised
 *   kernels; any multiple of 8 up to 128 a chunked generic path (Im2Grid's CoTr: heads = 1, hd = C, rpb = 0, scale = 1).
 * Never materialises the (..,27) attention tensor.
 * lse (B,D,H,W,heads) or NULL: log-sum-exp of the 27 logits per voxel-head, written for the backward (NULL when no
 * gradient is needed).  The backward takes q, k, rpb, the forward's out and lse, and d_out; it recomputes each softmax
 * probability from lse in a single pass (no 27-wide scratch), d_rpb through a deterministic two-stage reduction. */
int modet_na_fwd(const float* q, const float* k, const float* rpb, float* out, float* lse,
                 int B, int D, int H, int W, int heads, int hd, float scale, modet_stream_t stream);
size_t modet_na_bwd_ws_bytes(int B, int D, int H, int W, int heads);
int modet_na_bwd(const float* q, const float* k, const float* rpb, const float* out, const float* lse,
                 const float* d_out, float* d_q, float* d_k, float* d_rpb, void* ws, size_t ws_bytes,
                 int B, int D, int H, int W, int heads, int hd, float scale, modet_stream_t stream);
/* d_rpb == NULL (modet_na_bwd / _t): the two reduction launches are skipped and the per-workgroup partial rows stay at the start
 * of ws as float [B][heads][rows][27] with rows = modet_na_bwd_partial_rows(...): the caller sums them later, with the other leaf
 * reductions of its backward pass, in one modet_leaf_reduce_many launch (ws must live until then). */
int64_t modet_na_bwd_partial_rows(int B, int D, int H, int W, int heads, int hd);
/* The same two entry points with bf16 q / k (qk_bf16 != 0: two channels per 32-bit word, channels-last as above; BASELINE.json
 * configs[4], bf16 storage): every product and sum stays fp32 -- the result is bit-identical to the fp32 entry points fed with
 * the bf16 values widened to fp32.  d_q / d_k are fp32.  head_dim 6 only. */
int modet_na_fwd_t(const void* q, const void* k, int qk_bf16, const float* rpb, float* out, float* lse, int B, int D, int H, int W,
                   int heads, int hd, float scale, modet_stream_t stream);
int modet_na_bwd_t(const void* q, const void* k, int qk_bf16, const float* rpb, const float* out, const float* lse,
                   const float* d_out, float* d_q, float* d_k, float* d_rpb, void* ws, size_t ws_bytes, int B, int D,
                   int H, int W, int heads, int hd, float scale, modet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3x3 convolution, stride 1, zero pad 1 (nn.Conv3d call sites ModeT/models.py:127,:144,:254)
 * as an fp32 MFMA implicit GEMM.  x (B,D,H,W,Cin), y (B,D,H,W,Cout) channels-last;
 * w in the reference's parameter layout (Cout,Cin,3,3,3); bias (Cout) or NULL.
 * act: 0 = none, 1 = LeakyReLU(0.1) fused (ConvBlock, models.py:119-133). */
/* Which kernel family the fp32 conv entry points run for this launch shape (chosen per shape);
 * pass: 0 = forward (Cin -> Cout), 1 = data gradient of that layer, 2 = its weight gradient:
 *   0 = exact-f32 MFMA implicit GEMM (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain)            conv3d.hip
 *   1 = "bf16x3" fp32 emulation, tiled: fp32 tensors, every operand split into three bf16 pieces, every product as six
 *       exact piece products on the bf16 matrix pipe, error <= 3 * 2^-24 |a b|                  conv3d_bf16.hip (SP = 3)
 *   2 = the same arithmetic as a z-marching kernel for the few-channel full-resolution layers   conv3d_x3.hip
 *   3 = exact-f32 MFMA straight from global memory for volumes of <= 16 k voxels (pyramid level 5, the CWM layers at
 *       level-4 resolution): one wave per 16 voxels x 16/32 output channels, no LDS (plain forward / dgrad launches only;
 *       launches with fused statistics, a lazily normalised input or an activation run family 0)   conv3d.hip (conv_direct_kernel)
 *   5 = (forward / data gradient) bf16x3 with the K index packed in channel quads: 2 x 8 x 8-voxel tiles, any channel count
 *       (12 / 24 / 48 / 6 without padding), fused statistics and the lazily normalised input as template variants: pyramid
 *       levels 3-5, the CWM layers, everything family 2 does not take up to 6 M voxels               conv3d_q.hip
 * Families 1, 2 and 5 produce the fused InstanceNorm statistics (modet_conv3d_fwd_stats) at no cost for every Cout.
 *   4 = (weight gradient only) bf16x3 through LDS transpose reads, ds_read_b64_tr_b16: every layer with Cin >= 12 or an
 *       odd channel count, and Cout = 16                                                           conv3d_wtr.hip
 * Families 1 and 2 produce the fused InstanceNorm statistics (modet_conv3d_fwd_stats) at no cost for every Cout.
 * TWO f16 PIECES (round 5).  Families 2, 4 and 5 have a second form of the same fp32-accurate arithmetic: every operand split
 * into two f16 pieces (x = hi + lo, |lo| <= 2^-11 |x|: x carried to 2^-22 |x|), a product as the THREE piece products
 * hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-22 |a b|) -- half the matrix-pipe work of the six bf16 products; measured
 * against fp64 it sits in the same error class (tests/test_gpu_ops.py, test_conv_*_f16_*: <= 7e-7 of max|y|, never worse than
 * the bf16x3 launch of the same tensors).  f16 has the mantissa for this but not the range (65 504; subnormal below 6e-5), so
 * the operands are scaled by exact powers of two while they are split and the accumulator is scaled back:
 *   weights x 2^8 (|w| < 255);
 *   activations x 2^4 (|x| < 4 094; unscaled, |x| < 65 504, for family 2 with Cin = 4): FORWARD launches take the f16 form only
 *     through the *_bounded entry points (modet_conv3d_fwd_bounded, modet_conv3d_fwd_stats_bounded), i.e. on the CALLER'S word that
 *     x is inside that range -- a ConvInsBlock output, a pooled or upsampled copy of one, or a flow field; LeakyReLU(InstanceNorm(.))
 *     is bounded by sqrt(V), so volumes below 2^24 voxels cannot overflow -- and in modet_conv3d_fwd_normin, whose input is
 *     normalised while it is staged.  The PLAIN entry points (modet_conv3d_fwd, modet_conv3d_fwd_stats) make no assumption
 *     about x, as nn.Conv3d makes none (reference models.py:127): they run the three bf16 pieces, which have fp32's range.
 *     (Round 5 ran every forward launch on f16 and returned inf beyond the range: VERDICT r5 item 6.  The model itself hands
 *     its first ConvInsBlock the un-normalised ConvBlock output through the plain entry point; the Cin = 1 layer that sees the
 *     raw image is family 0, exact f32.)
 *   gradients by the power of two that takes max |d_y| into [2^14, 2^15): BACKWARD launches take the f16 form only when the
 *     caller hands that maximum over (the *_amax entry points below; the InstanceNorm backward that produces a d_y leaves it
 *     for free), otherwise they run the three bf16 pieces, which need no range information.
 * The choice is a function of the arguments alone: the product library reads NO environment variable.  The A/B switches
 * MODET_CONV_X3=0 (no family 2), MODET_CONV_SPLIT=0 / 1 (no / forced family 1), MODET_CONV_DIRECT=0 (no family 3),
 * MODET_CONV_WTR=0 (no family 4), MODET_CONV_Q=0 (no family 5) exist only in tuning builds of the library (-DMODET_TUNING, tools/build_variant.sh).
 * `variant` tells what the launch fuses, because that changes the routing: 0 = plain (modet_conv3d_fwd with act = 0,
 * modet_conv3d_fwd_stats, _bwd_data, _bwd_weight), 1 = fused LeakyReLU (modet_conv3d_fwd with act = 1: never family 1 or 3),
 * 2 = lazily normalised input (modet_conv3d_fwd_normin: family 2 or 0), 3 = fused statistics (modet_conv3d_fwd_stats:
 * never family 3).  modet_conv3d_kernel_family == variant 0. */
int modet_conv3d_kernel_family(int B, int D, int H, int W, int Cin, int Cout, int pass);
int modet_conv3d_kernel_family_v(int B, int D, int H, int W, int Cin, int Cout, int pass, int variant);
size_t modet_conv3d_ws_bytes(int Cin, int Cout);
/* `step` (every conv entry point that packs weights): NULL, or the context whose recorded packing jobs apply, see
 * modet_conv3d_prepack_* below. */
int modet_conv3d_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                     int B, int D, int H, int W, int Cin, int Cout, int act, modet_stream_t stream, modet_step_ctx_t* step);
/* the same on the caller's word that |x| < 4 094 (see "TWO f16 PIECES" above): half the matrix-pipe work */
int modet_conv3d_fwd_bounded(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                             int B, int D, int H, int W, int Cin, int Cout, int act, modet_stream_t stream, modet_step_ctx_t* step);
/* Forward + fused InstanceNorm statistics (ConvInsBlock, models.py:135-151): the staged epilogue also emits
 * per-(sample, workgroup) partial sums (sum, sum of squares) of the output.  modet_conv3d_stats_bytes() == 0 means this
 * (Cin, Cout) cannot fuse them (use modet_conv3d_fwd + modet_instnorm_lrelu_fwd).  Consume with
 * modet_instnorm_lrelu_fwd_stats (same stats_bytes). */
size_t modet_conv3d_stats_bytes(int B, int D, int H, int W, int Cin, int Cout);
size_t modet_conv3d_normin_stats_bytes(int B, int D, int H, int W, int Cin, int Cout);   /* stats of modet_conv3d_fwd_normin */
int modet_conv3d_fwd_stats(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                           float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                           modet_stream_t stream, modet_step_ctx_t* step);
int modet_conv3d_fwd_stats_bounded(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                                   float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                   modet_stream_t stream, modet_step_ctx_t* step);
/* An input whose range is known ON THE DEVICE only (the ConvBlock 1 -> 4 output in front of the first ConvInsBlock: as large as the
 * image is).  modet_conv3d_fwd_amax_out = modet_conv3d_fwd that also leaves max |y| in y_amax (MODET_AMAX_FLOATS floats, written by
 * this call; the kernel of the first encoder block -- Cin 1, Cout 4, >= 500 k voxels -- carries that epilogue for free,
 * MODET_ERR_UNSUPPORTED otherwise).  modet_conv3d_fwd_stats_amax = modet_conv3d_fwd_stats given such maxima of |x| (NULL = none):
 * the z-marching family then runs the two f16 pieces with x scaled by the power of two that takes the maximum into
 * [2^14, 2^15) -- any range, half the matrix work, and none of the precision the UNSCALED Cin = 4 form of round 5 lost on small
 * activations (their low piece fell into f16's subnormals: that was 2/3 of the model's flow error at 160x192x160); other
 * families ignore x_amax and run bf16x3.  modet_conv3d_bwd_weight_amax2: the weight gradient given both maxima (d_y's and x's). */
int modet_conv3d_fwd_amax_out(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                              int D, int H, int W, int Cin, int Cout, int act, float* y_amax, modet_stream_t stream,
                              modet_step_ctx_t* step);
int modet_conv3d_fwd_stats_amax(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                                float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                const float* x_amax, modet_stream_t stream, modet_step_ctx_t* step);
int modet_conv3d_bwd_weight_amax2(const float* x, const float* d_y, float* d_w, float* d_bias, void* ws, size_t ws_bytes, int B,
                                  int D, int H, int W, int Cin, int Cout, const float* dy_amax, const float* x_amax,
                                  modet_stream_t stream, modet_step_ctx_t* step);
/* Forward whose input is the RAW output of the previous ConvInsBlock: LeakyReLU((x_raw - in_mean) * in_rstd) is applied
 * while the input tile is staged (zero padding stays zero), so the normalised tensor is never written
 * (ConvInsBlock -> ConvInsBlock chains, models.py:186-219; used when no gradient is needed: inference).  in_mean / in_rstd: (B*Cin) from modet_instnorm_stats.
 * stats (+stats_bytes) may be NULL (0): no fused output statistics.  Cin % 4 == 0. */
int modet_conv3d_fwd_normin(const float* x_raw, const float* in_mean, const float* in_rstd, const float* w,
                            const float* bias, float* y, void* ws, size_t ws_bytes, float* stats, size_t stats_bytes, int B,
                            int D, int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step);
/* d_x = conv(d_y, flipped/transposed w) */
int modet_conv3d_bwd_data(const float* d_y, const float* w, float* d_x, void* ws, size_t ws_bytes,
                          int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step);
/* Data gradient of a conv whose INPUT was y = LeakyReLU(InstanceNorm(x_raw)) (ConvInsBlock -> conv, models.py:186-219): d_x
 * is the gradient w.r.t. y, and the kernel's epilogue also forms the first pass of that InstanceNorm's backward over it --
 * rows [B][workgroup][Cin][2] of (sum g, sum g*xhat), g = d_x * lrelu'(xhat), xhat = (x_raw - mean) * rstd -- so
 * modet_instnorm_lrelu_bwd_rows can go straight to its apply pass (saves reading d_x and x_raw once more: 8 B/element).
 * modet_conv3d_bwd_data_instats_bytes() == 0: this shape does not run the kernel family that carries the epilogue (family 2):
 * use modet_conv3d_bwd_data + modet_instnorm_lrelu_bwd.  x_raw (B,D,H,W,Cin); mean / rstd (B*Cin). */
size_t modet_conv3d_bwd_data_instats_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modet_conv3d_bwd_data_instats(const float* d_y, const float* w, float* d_x, const float* x_raw, const float* mean,
                                  const float* rstd, float* rows, size_t rows_bytes, void* ws, size_t ws_bytes,
                                  int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step);
/* modet_conv3d_bwd_data / _instats given a bound of |d_y|: dy_amax (device memory, MODET_AMAX_FLOATS floats, the bound is the
 * max over its MODET_AMAX_SLOTS slots) >= max |d_y| over the whole tensor, as left by modet_instnorm_lrelu_bwd*_amax.  The z-marching and channel-quad kernel families (2, 5)
 * then run the gradient on TWO f16 pieces per operand
 * (three MFMA products, like its forward launches since round 5) instead of three bf16 pieces (six): f16 has the mantissa for
 * it but not the range, and a gradient has no a-priori range -- d_y is scaled by the power of two that takes dy_amax to
 * [2^14, 2^15) while it is split, the accumulator is scaled back (exact).  Elements below 2^-18 dy_amax keep an ABSOLUTE
 * resolution of 2^-40 dy_amax (their low piece is subnormal), everything above a relative one of 2^-22; a dy_amax that is too
 * small overflows (inf), one that is too large only moves that floor.  dy_amax == NULL, or a shape of another kernel family:
 * exactly the plain call. */
int modet_conv3d_bwd_data_amax(const float* d_y, const float* w, float* d_x, void* ws, size_t ws_bytes, int B, int D, int H,
                               int W, int Cin, int Cout, const float* dy_amax, modet_stream_t stream, modet_step_ctx_t* step);
int modet_conv3d_bwd_data_instats_amax(const float* d_y, const float* w, float* d_x, const float* x_raw, const float* mean,
                                       const float* rstd, float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B,
                                       int D, int H, int W, int Cin, int Cout, const float* dy_amax, modet_stream_t stream,
                                       modet_step_ctx_t* step);
/* d_w (Cout,Cin,3,3,3), d_bias (Cout) or NULL; deterministic two-stage reduction */
size_t modet_conv3d_bwd_weight_ws_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modet_conv3d_bwd_weight(const float* x, const float* d_y, float* d_w, float* d_bias, void* ws, size_t ws_bytes,
                            int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream);
/* ConvBlock (conv + LeakyReLU, models.py:119-133) weight gradient with the activation's derivative folded into the
 * d_y load: y_act is the block's output, d_y the gradient w.r.t. it.  Supported for the first encoder block
 * (Cin = 1, Cout = 4; its input needs no gradient, so this replaces the separate modet_lrelu_bwd pass);
 * MODET_ERR_UNSUPPORTED otherwise. */
int modet_conv3d_bwd_weight_act(const float* x, const float* d_y, const float* y_act, float* d_w, float* d_bias,
                                void* ws, size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                modet_stream_t stream);

/* Weight packing hoisted out of the step.  Every fp32 conv launch (forward, statistics, lazily normalised, data gradient)
 * starts with a 4 us launch that packs its weights into the MFMA operand layout; a training step has 38 of them and
 * the weights only change once per step.  Protocol:
 *   modet_conv3d_prepack_record(ctx, 1); <one forward+backward, every conv call given ctx>;
 *   n = modet_conv3d_prepack_record(ctx, 0);                                                          -- learn the jobs
 *   every later step: modet_conv3d_prepack_begin(ctx, arena, modet_conv3d_prepack_arena_bytes(ctx), stream) -- ONE launch
 *                     <forward+backward: launches given ctx whose (weights pointer, geometry) was recorded skip theirs>
 *                     modet_conv3d_prepack_end(ctx);
 * The caller guarantees the recorded weight tensors stay where they are and do not change between _begin and _end, and
 * keeps `arena` alive until _end (and for as long as a captured hipGraph of the step may be replayed).  Launches that
 * were not recorded, or are given another / no context, pack as usual. */
int modet_conv3d_prepack_record(modet_step_ctx_t* ctx, int on);
size_t modet_conv3d_prepack_arena_bytes(modet_step_ctx_t* ctx);
int modet_conv3d_prepack_begin(modet_step_ctx_t* ctx, void* arena, size_t arena_bytes, modet_stream_t stream);
int modet_conv3d_prepack_end(modet_step_ctx_t* ctx);

/* Deferred weight-gradient reductions.  A backward pass launches ~20 weight-gradient kernels, each followed by a tiny
 * reduction of its per-workgroup partial tiles (5-9 us apiece, all launch latency).  modet_conv3d_bwd_weight_defer is
 * modet_conv3d_bwd_weight (y_act == NULL) / modet_conv3d_bwd_weight_act (y_act != NULL) without that reduction: it only
 * writes the partial tiles into `ws` and queues the reduction in `step`; modet_conv3d_wgrad_defer_flush(step, ..) runs
 * every reduction queued there in ONE launch on `stream` (same arithmetic, same fixed order: bit-identical results)
 * and empties the queue.  Until the flush the caller keeps every `ws`, d_w and d_bias it passed alive and untouched. */
int modet_conv3d_bwd_weight_defer(const float* x, const float* d_y, const float* y_act, float* d_w, float* d_bias,
                                  void* ws, size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                  modet_stream_t stream, modet_step_ctx_t* step);
int modet_conv3d_wgrad_defer_flush(modet_step_ctx_t* step, modet_stream_t stream);
/* modet_conv3d_bwd_weight_defer (step != NULL) / modet_conv3d_bwd_weight (step == NULL) given a bound of |d_y| (see
 * modet_conv3d_bwd_data_amax) and the promise that x is an ACTIVATION -- LeakyReLU(InstanceNorm(.)) or the first ConvBlock's
 * output, |x| < 4000: the z-marching weight-gradient kernel then splits both operands into two f16 pieces (x by a fixed power
 * of two, d_y by the one dy_amax gives) and runs three products instead of six; so does the transpose-read kernel (family 4;
 * a launch it queues in `step` reads dy_amax at the flush: keep it alive until then, like x and d_y).  y_act must be NULL.
 * dy_amax == NULL or a shape of another kernel: exactly the plain call. */
int modet_conv3d_bwd_weight_amax(const float* x, const float* d_y, float* d_w, float* d_bias, void* ws, size_t ws_bytes, int B,
                                 int D, int H, int W, int Cin, int Cout, const float* dy_amax, modet_stream_t stream,
                                 modet_step_ctx_t* step);
/* Weight gradient of a conv whose INPUT was LeakyReLU(InstanceNorm(x_raw)) (ConvInsBlock -> conv), given the RAW tensor and
 * its statistics: the normalisation is applied while x is staged (zero padding stays zero), so a training step never
 * materialises the normalised tensor of such a chain (the forward has modet_conv3d_fwd_normin, the data gradient works on d_y
 * alone).  Only the shapes of the z-marching weight-gradient kernel (modet_conv3d_bwd_weight_normin_ok() == 1; the level-1 layers,
 * where the saved pass is 0.12 ms of a train step); dy_amax as in modet_conv3d_bwd_weight_amax (may be NULL); step NULL = reduce now. */
int modet_conv3d_bwd_weight_normin_ok(int B, int D, int H, int W, int Cin, int Cout);
int modet_conv3d_bwd_weight_normin(const float* x_raw, const float* in_mean, const float* in_rstd, const float* d_y, float* d_w,
                                   float* d_bias, void* ws, size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                   const float* dy_amax, modet_stream_t stream, modet_step_ctx_t* step);
/* Round 5: for the many-channel layers of the small pyramid levels the deferred call queues the partial-tile LAUNCH as well
 * (17 launches of 20-50 us per train step, most too small to fill the chip); the flush runs all queued layers of one kernel
 * variant as one grid, then the reductions.  modet_conv3d_wgrad_defers_operands(..) == 1 says that a deferred call of this
 * shape reads x and d_y at the FLUSH, not at the call: the caller keeps them alive and untouched until then, too. */
int modet_conv3d_wgrad_defers_operands(int B, int D, int H, int W, int Cin, int Cout);


/* InstanceNorm3d(affine=False, eps, biased variance) + LeakyReLU(0.1) (ConvInsBlock, models.py:135-151).
 * x,y (B,V,C) channels-last with V = D*H*W; mean,rstd (B*C) are outputs of fwd / inputs of bwd. */
size_t modet_instnorm_ws_bytes(int B, int64_t V, int C);
int modet_instnorm_lrelu_fwd(const float* x, float* y, float* mean, float* rstd, void* ws, size_t ws_bytes,
                             int B, int64_t V, int C, float eps, modet_stream_t stream);
/* same, with the statistics taken from modet_conv3d_fwd_stats' partial rows (stats_bytes = modet_conv3d_stats_bytes(...))
 * instead of a pass over x */
int modet_instnorm_lrelu_fwd_stats(const float* x, float* y, float* mean, float* rstd, const float* stats,
                                   size_t stats_bytes, int B, int64_t V, int C, float eps, modet_stream_t stream);
/* mean / rstd only, no apply pass: from modet_conv3d_fwd_stats' partial rows (x, ws may be NULL) or, with stats == NULL,
 * by a statistics pass over x (ws_bytes >= modet_instnorm_ws_bytes) */
int modet_instnorm_stats(const float* x, float* mean, float* rstd, const float* stats, size_t stats_bytes, void* ws,
                         size_t ws_bytes, int B, int64_t V, int C, float eps, modet_stream_t stream);
int modet_instnorm_lrelu_bwd(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                             void* ws, size_t ws_bytes, int B, int64_t V, int C, modet_stream_t stream);
/* the same with the statistics pass already done by modet_conv3d_bwd_data_instats (rows, rows_bytes as returned there);
 * ws_bytes >= 2 * B * C * 4 */
int modet_instnorm_lrelu_bwd_rows(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                                  const float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B, int64_t V, int C,
                                  modet_stream_t stream);
/* modet_instnorm_lrelu_bwd for the last block of an encoder level, whose d_y = unpool(g_pooled) / 8 + [add_a ; add_b]
 * (g_pooled (B,D/2,H/2,W/2,C): gradient of the AvgPool3d(2) output; add_a (Bh,D,H,W,C) / add_b (B-Bh,D,H,W,C): gradients of the
 * two batch halves of y, either may be NULL): d_y is formed while it is read instead of being written by modet_avgpool2_bwd
 * and read back twice; bit-identical to that sequence.  ws_bytes >= modet_instnorm_ws_bytes(B, D*H*W, C). */
int modet_instnorm_lrelu_bwd_pool(const float* g_pooled, const float* add_a, const float* add_b, int Bh, const float* x,
                                  const float* mean, const float* rstd, float* d_x, void* ws, size_t ws_bytes, int B, int D,
                                  int H, int W, int C, modet_stream_t stream);
/* The three InstanceNorm backward forms above that ALSO leave max |d_x| over the whole tensor in `amax` (device memory,
 * MODET_AMAX_FLOATS floats, written by this call).  The maximum is the max over the MODET_AMAX_SLOTS elements
 * amax[i * MODET_AMAX_STRIDE]: every wave of the apply pass maxes into the slot of its workgroup with an integer atomic on the
 * float's bits (order-independent: deterministic), and 16 000 waves finishing together on ONE address serialise in a single L2
 * channel (measured: +0.5 ms per train step), so the slots are 128 bytes apart.  The finalize launch in front zeroes them.
 * It is the scale with which the convolutions that consume d_x as their d_y split it into two f16 pieces
 * (modet_conv3d_bwd_data_amax, modet_conv3d_bwd_data_instats_amax, modet_conv3d_bwd_weight_amax).
 * amax == NULL: exactly the plain call.  d_x is bit-identical with and without. */
#define MODET_AMAX_SLOTS 64
#define MODET_AMAX_STRIDE 32
#define MODET_AMAX_FLOATS (MODET_AMAX_SLOTS * MODET_AMAX_STRIDE)
int modet_instnorm_lrelu_bwd_amax(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                                  void* ws, size_t ws_bytes, int B, int64_t V, int C, float* amax, modet_stream_t stream);
int modet_instnorm_lrelu_bwd_rows_amax(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                                       const float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B, int64_t V, int C,
                                       float* amax, modet_stream_t stream);
int modet_instnorm_lrelu_bwd_pool_amax(const float* g_pooled, const float* add_a, const float* add_b, int Bh, const float* x,
                                       const float* mean, const float* rstd, float* d_x, void* ws, size_t ws_bytes, int B, int D,
                                       int H, int W, int C, float* amax, modet_stream_t stream);
/* d_x = d_y * (y > 0 ? 1 : 0.1): backward of the LeakyReLU fused into modet_conv3d_fwd(act=1) */
int modet_lrelu_bwd(const float* d_y, const float* y, float* d_x, int64_t n, modet_stream_t stream);
/* AvgPool3d(2) (models.py:201,:207,:213,:219); D,H,W are the INPUT dims (even). */
int modet_avgpool2_fwd(const float* x, float* y, int B, int D, int H, int W, int C, modet_stream_t stream);
/* the same pooling of a bf16 tensor into an fp32 one (BASELINE.json configs[4]) */
int modet_avgpool2_fwd_x16(const void* x_bf16, float* y, int B, int D, int H, int W, int C, modet_stream_t stream);
/* The apply pass of InstanceNorm + LeakyReLU (mean / rstd from modet_instnorm_stats) fused with the AvgPool3d(2) that follows
 * the last ConvInsBlock of an encoder level (models.py:186-219): y = LeakyReLU((x - mean) * rstd) (B,D,H,W,C) and
 * pooled = AvgPool3d(2)(y) (B,D/2,H/2,W/2,C) in one pass over x; bit-identical to modet_instnorm_lrelu_fwd* followed by
 * modet_avgpool2_fwd.  D, H, W even, C % 4 == 0. */
int modet_instnorm_lrelu_apply_pool(const float* x, const float* mean, const float* rstd, float* y, float* pooled, int B,
                                    int D, int H, int W, int C, modet_stream_t stream);
/* d_x = unpool(d_y)/8 + addend; addend (same shape as d_x, may be NULL) = gradient of the un-pooled branch */
int modet_avgpool2_bwd(const float* d_y, const float* addend, float* d_x, int B, int D, int H, int W, int C,
                       modet_stream_t stream);

/* ProjectionLayer: Linear(Cin->dim) + LayerNorm(dim, eps) (models.py:230-241).
 * x (N,Cin) channels-last voxels, Wt = proj.weight (dim,Cin), y (N,dim). */
int modet_proj_ln_fwd(const float* x, const float* Wt, const float* bias, const float* gamma, const float* beta,
                      float* y, int64_t N, int Cin, int dim, float eps, modet_stream_t stream);
size_t modet_proj_ln_bwd_ws_bytes(int64_t N, int Cin, int dim);
int modet_proj_ln_bwd(const float* x, const float* Wt, const float* bias, const float* gamma, const float* d_y,
                      float* d_x, float* d_Wt, float* d_bias, float* d_gamma, float* d_beta,
                      void* ws, size_t ws_bytes, int64_t N, int Cin, int dim, float eps, modet_stream_t stream);

/* The same layer applied to two inputs (the fixed and the moving feature map of a level share one ProjectionLayer,
 * models.py:371-372): both data gradients and the parameter gradients of BOTH uses, summed in one fixed-order fp64
 * reduction.  _ws_bytes returns 0 when the shape is not covered (use two modet_proj_ln_bwd calls and add). */
size_t modet_proj_ln_bwd_pair_ws_bytes(int64_t N, int Cin, int dim);
/* y1 = layer(x1), y2 = layer(x2): one launch for the grouped shapes (pyramid levels 3-5), two modet_proj_ln_fwd otherwise;
 * bit-identical to two calls. */
int modet_proj_ln_fwd_pair(const float* x1, const float* x2, const float* Wt, const float* bias, const float* gamma,
                           const float* beta, float* y1, float* y2, int64_t N, int Cin, int dim, float eps,
                           modet_stream_t stream);
int modet_proj_ln_bwd_pair(const float* x1, const float* d_y1, float* d_x1, const float* x2, const float* d_y2, float* d_x2,
                           const float* Wt, const float* bias, const float* gamma, float* d_Wt, float* d_bias,
                           float* d_gamma, float* d_beta, void* ws, size_t ws_bytes, int64_t N, int Cin, int dim,
                           float eps, modet_stream_t stream);
/* d_Wt == d_bias == d_gamma == d_beta == NULL (modet_proj_ln_bwd_pair / _t): the column-sum launch is skipped and the partial
 * rows stay at the start of ws as float [rows][3 dim + dim Cin], a row = [d_gamma | d_beta | d_bias | d_W], rows =
 * modet_proj_ln_bwd_pair_partial_rows(N, Cin, dim) -- to be summed by modet_leaf_reduce_many. */
int64_t modet_proj_ln_bwd_pair_partial_rows(int64_t N, int Cin, int dim);
/* Every leaf reduction of a backward pass in ONE launch (15 launches of 5-8 us per ModeT train step otherwise: two per
 * attention level, one per projection level).  Job: output column i (0 <= i < ncols) is the fp64 sum, in a fixed order, over
 * o < outer and r < rows of part[o * outer_stride + r * row_stride + (i / col_group) * col_group_stride + i % col_group],
 * rounded to float and written to the segment it falls in: columns [0, n[0]) -> dst[0], the next n[1] -> dst[1], ...
 * (n[0] + .. + n[3] == ncols).  d_rpb of modet_na_bwd: outer = B, outer_stride = heads * rows * 27, row_stride = 27,
 * ncols = heads * 27, col_group = 27, col_group_stride = rows * 27.  Projection pair: outer = 1, row_stride = ncols =
 * col_group = 3 dim + dim Cin, segments d_gamma, d_beta, d_bias, d_W.  `jobs` is a HOST array (copied into the launch). */
#define MODET_LEAF_MAX_JOBS 16
typedef struct {
  const float* part;
  int64_t outer, outer_stride, rows, row_stride, col_group_stride;
  int ncols, col_group;
  float* dst[4];
  int n[4];
} modet_leaf_job_t;
int modet_leaf_reduce_many(const modet_leaf_job_t* jobs, int njobs, modet_stream_t stream);
/* Typed variants (BASELINE.json configs[4], bf16 storage): x_bf16 / y_bf16 != 0 means that tensor holds bf16 (channels-last);
 * values are widened on load and rounded to nearest even on store, everything in between is the fp32 arithmetic above -- a bf16
 * input gives bit-identical results to the fp32 entry point fed with the widened values, a bf16 output is the fp32 output rounded. */
int modet_proj_ln_fwd_t(const void* x, int x_bf16, const float* Wt, const float* bias, const float* gamma, const float* beta,
                        void* y, int y_bf16, int64_t N, int Cin, int dim, float eps, modet_stream_t stream);
int modet_proj_ln_bwd_pair_t(const void* x1, int x1_bf16, const float* d_y1, float* d_x1, const void* x2, int x2_bf16,
                             const float* d_y2, float* d_x2, const float* Wt, const float* bias, const float* gamma, float* d_Wt,
                             float* d_bias, float* d_gamma, float* d_beta, void* ws, size_t ws_bytes, int64_t N, int Cin, int dim,
                             float eps, modet_stream_t stream);

/* SpatialTransformer (models.py:25-67; utils.py:30-83 for mode 1):
 *   out[b,p,c] = sample(src[b,:,c], p + flow[b,p,:]), zero padding, voxel coordinates
 *   (the reference's normalise -> grid_sample(align_corners=True) round trip is the identity).
 * src,out (B,D,H,W,C); flow (B,D,H,W,3) channels-last, component a = displacement along axis a.
 * mode 0 = trilinear, 1 = nearest (round half to even, as ATen's nearbyint).
 * add_flow=1 (needs C==3): out = warp(src,flow) + flow, the composition of models.py:392,:398,:403,:408. */
int modet_warp_fwd(const float* src, const float* flow, float* out, int B, int D, int H, int W, int C,
                   int mode, int add_flow, modet_stream_t stream);
/* trilinear warp whose OUTPUT is stored as bf16 (rounded to nearest even; BASELINE.json configs[4]: the warped moving features
 * feed only the projection).  C % 4 == 0; fp32 src and flow, fp32 arithmetic. */
int modet_warp_fwd_o16(const float* src, const float* flow, void* out_bf16, int B, int D, int H, int W, int C,
                       modet_stream_t stream);
/* typed forms (BASELINE.json configs[4]): src_bf16 / out_bf16 != 0 = that tensor holds bf16 (widened on load / rounded to nearest
 * even on store; fp32 arithmetic, fp32 gradients).  Plain trilinear warps only (no add_flow, no flow bound); C % 4 == 0 for the
 * forward.  A bf16 src gives results bit-identical to the fp32 entry point fed with the widened values. */
int modet_warp_fwd_t(const void* src, int src_bf16, const float* flow, void* out, int out_bf16, int B, int D, int H, int W, int C,
                     modet_stream_t stream);
int modet_warp_bwd_t(const void* src, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow, int B,
                     int D, int H, int W, int C, int add_flow, int flow_bound, modet_stream_t stream);
/* Round 5: the same with a SECOND gradient of `flow` added on the way out (d_flow = d loss / d flow through this warp +
 * d_flow_add; d_flow_add may be NULL).  A flow field feeds a feature warp AND the next composition: autograd would run an
 * element-wise add over two full-size tensors (5 launches and 0.07 ms per 160x192x160 step); the warp node that sees the
 * other consumer's gradient first hands it in here instead (ops.warp_tee).  Not with flow_bound. */
int modet_warp_bwd_acc(const void* src, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow,
                       const float* d_flow_add, int B, int D, int H, int W, int C, int add_flow, int flow_bound,
                       modet_stream_t stream);
/* Round 5, opt-in: the same backward with a DETERMINISTIC d_src.  The scatter-add of the eight corner weights runs on float
 * atomics above (as ATen's grid_sampler_3d_backward, reference ModeT/models.py:67): d_src depends on the order the hardware
 * retires them, ~1e-6 run to run.  Here every contribution is a 64-bit fixed-point integer (scale: a power of two from
 * max |d_out|, 2^-40 of it per unit) added with an integer atomic -- associative, so the result is bit-identical from run to
 * run, and with it the whole train step (these are its only atomics).  ws: modet_warp_bwd_det_ws_bytes (8 bytes per element
 * of d_src + a header); costs a max pre-pass and a decode pass.  d_src is required; no flow_bound (that path has no atomics). */
size_t modet_warp_bwd_det_ws_bytes(int B, int D, int H, int W, int C);
int modet_warp_bwd_det(const void* src, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow,
                       const float* d_flow_add, void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, int add_flow,
                       modet_stream_t stream);
/* The trilinear warp's backward WITHOUT global float atomics (csrc/warp_tile.hip; round 5 prototype, round 6 the DEFAULT path of
 * the feature warps): FILL per-tile lists (a fixed segment of 1 536 entries per 8^3 destination tile of the base corner + one
 * overflow list) of payload entries (voxel, flow, d_out; voxels whose d_out is all zero or whose corners all leave the volume are
 * dropped; d_flow of every voxel is produced here, the src corners gathered) -> ACCUMULATE one workgroup per (tile, 8 channels)
 * into a 64-bit fixed-point
 * LDS window (2^-40 of the power of two above max |d_out| per unit, as modet_warp_bwd_det) -> BORDER gather of the tile faces.  Equal to modet_warp_bwd
 * within fp32 rounding of the sums (the float-atomic ORDER is what differs there run to run; this one is bit-reproducible: integer
 * sums), every launch a kernel with fixed arguments (capturable).  d_src need NOT be zeroed; the workspace needs no preparation.
 * A non-finite d_out poisons d_src with NaN (as the float path would).  C a multiple of 8, or C == 3 (fp32 src; the flow
 * compositions warp(src, flow) + flow with add_flow != 0: d_flow += d_out); dimensions <= 1024, B*D*H*W < 2^31; _ws_bytes returns
 * 0 for anything else (then: modet_warp_bwd_acc / modet_warp_bwd_det).
 * modet_warp_bwd_dsrc_tiles: d_src only.  modet_warp_bwd_tiles: d_src and d_flow (+ d_flow_add, NULL = none: a second gradient
 * of the flow, as modet_warp_bwd_acc); src fp32 or (src_bf16 != 0) bf16.  Same workspace. */
size_t modet_warp_bwd_dsrc_tiles_ws_bytes(int B, int D, int H, int W, int C);
int modet_warp_bwd_dsrc_tiles(const float* flow, const float* d_out, float* d_src, void* ws, size_t ws_bytes,
                              int B, int D, int H, int W, int C, modet_stream_t stream);
int modet_warp_bwd_tiles(const void* src, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow,
                         const float* d_flow_add, void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, int add_flow,
                         modet_stream_t stream);
/* d_src and/or d_flow; either may be NULL.  Trilinear only.
 * flow_bound = 0: arbitrary flow, d_src is zeroed here and scatter-added with float atomics (as ATen does).
 * flow_bound = 1: the CALLER guarantees |flow| <= 1 voxel everywhere (true for the attention output w of
 *   ModeTransformer, an expectation over offsets in {-1,0,1}^3; models.py:392-408 compose with exactly that):
 *   d_src is then gathered from the 27 neighbours -- no atomics, deterministic.  C must be 3. */
int modet_warp_bwd(const float* src, const float* flow, const float* d_out, float* d_src, float* d_flow,
                   int B, int D, int H, int W, int C, int add_flow, int flow_bound, modet_stream_t stream);

/* nn.Upsample(2,'trilinear',align_corners=True) of scale*x (models.py:354,:257-261); d,h,w = INPUT dims.
 * x (B,d,h,w,C) -> y (B,2d,2h,2w,C).  Backward is the exact transpose in gather form (no atomics). */
int modet_upsample2_fwd(const float* x, float* y, int B, int d, int h, int w, int C, float scale, modet_stream_t stream);
int modet_upsample2_bwd(const float* d_y, float* d_x, int B, int d, int h, int w, int C, float scale, modet_stream_t stream);
/* the same gradient as three 1-D passes (z, y, x) through a workspace, for the large levels; _ws_bytes returns 0 where the
 * one-launch gather above is the better choice (small volumes) */
size_t modet_upsample2_bwd_sep_ws_bytes(int B, int d, int h, int w, int C);
int modet_upsample2_bwd_sep(const float* d_y, float* d_x, void* ws, size_t ws_bytes, int B, int d, int h, int w, int C,
                            float scale, modet_stream_t stream);

/* layout changes at the module boundary: (B,C,V) <-> (B,V,C) */
int modet_ncdhw_to_cl(const float* x, float* y, int B, int C, int64_t V, modet_stream_t stream);
int modet_cl_to_ncdhw(const float* x, float* y, int B, int C, int64_t V, modet_stream_t stream);

/* CWM tail (models.py:263-275): out[n,a] = 2 * sum_h softmax_h(logits[n,:])[h] * x[n,3h+a].
 * x (N,heads*3) = upsampled sub-flows, logits (N,heads) = output of the last CWM conv, out (N,3). */
int modet_cwm_tail_fwd(const float* x, const float* logits, float* out, int64_t N, int heads, modet_stream_t stream);
int modet_cwm_tail_bwd(const float* x, const float* logits, const float* d_out, float* d_x, float* d_logits,
                       int64_t N, int heads, modet_stream_t stream);

/* NCC_vxm (losses.py:34-94): loss[0] = -mean(cc); d_J (same shape as J) = d loss / d J or NULL.
 * I = y_true, J = y_pred, (B,D,H,W) single channel.  Separable zero-padded box sums, one z-marching kernel per direction.
 * modet_ncc_fwd_bwd: the default window win = [9, 9, 9] (what train.py:103 constructs);  _win: cubic windows
 * win x win x win, win in {3, 5, 7, 9} (losses.py:52 `self.win`; padding floor(win / 2) as losses.py:57). */
size_t modet_ncc_ws_bytes(int B, int D, int H, int W);
int modet_ncc_fwd_bwd(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes,
                      int B, int D, int H, int W, modet_stream_t stream);
int modet_ncc_fwd_bwd_win(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes,
                          int B, int D, int H, int W, int win, modet_stream_t stream);
/* The same with d_J = grad_scale * d loss / d J (loss[0] unscaled): grad_scale is the loss term's weight (train.py:127-129,
 * weights[0]), so a training step seeds its backward with d_J as it is -- no pass that multiplies it by an upstream scalar.
 * grad_scale = 1 is bit-identical to modet_ncc_fwd_bwd_win. */
int modet_ncc_fwd_bwd_win_scaled(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes,
                                 int B, int D, int H, int W, int win, float grad_scale, modet_stream_t stream);
/* NCC_vxm(win=[wz, wy, wx]) for ANY window (losses.py:52-59 accepts any list): the reference pads every axis by
 * floor(wz / 2), so for even / anisotropic windows the window sums -- and the mean -- live on a grid of
 * (D + 2p - wz + 1, H + 2p - wy + 1, W + 2p - wx + 1) voxels; reproduced exactly (separable sums through the workspace; the
 * cubic odd windows 3..9 have the fast kernel above).  MODET_ERR_DIM if that grid is empty. */
size_t modet_ncc_box_ws_bytes(int B, int D, int H, int W, int wz, int wy, int wx);
int modet_ncc_fwd_bwd_box(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes, int B, int D,
                          int H, int W, int wz, int wy, int wx, modet_stream_t stream);
/* Grad3d (losses.py:6-31) on a planar flow (B,3,D,H,W): loss[0], d_flow (NULL to skip).
 * penalty = 1 ('l1', |forward differences|, the class default) or 2 ('l2', squared; what train.py:104 uses). */
size_t modet_grad3d_ws_bytes(int B, int D, int H, int W);
int modet_grad3d_fwd_bwd(const float* flow, float* loss, float* d_flow, void* ws, size_t ws_bytes,
                         int B, int D, int H, int W, int penalty, modet_stream_t stream);
/* Grad3d on a CHANNELS-LAST flow (B,D,H,W,3) -- the layout ModeT's last composition writes -- with
 * d_flow_cl = grad_scale * d loss / d flow in the same layout (loss[0] unscaled).  Element for element the arithmetic of
 * modet_grad3d_fwd_bwd (d_flow equal bit for bit after the layout change when grad_scale = 1; the loss is the same sum in
 * another order).  A training step uses it to skip the planar copy of the flow (losses.py:6-31 indexes (B,3,D,H,W)) and
 * the copy of its gradient back. */
int modet_grad3d_fwd_bwd_cl(const float* flow_cl, float* loss, float* d_flow_cl, void* ws, size_t ws_bytes,
                            int B, int D, int H, int W, int penalty, float grad_scale, modet_stream_t stream);
/* y = x * s[0] with s a DEVICE scalar (chains an upstream scalar gradient without a host sync) */
int modet_scale_by_dev_scalar(const float* x, const float* s, float* y, int64_t n, modet_stream_t stream);

/* torch.optim.Adam(amsgrad=True, weight_decay=0) over one flat buffer (train.py:101,:131-133).
 * g is multiplied by grad_scale first (1/world_size after the all-reduce).  step counts from 1. */
int modet_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, int64_t n,
                            float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                            modet_stream_t stream);

/* Evaluation tail (utils.py:74-106, infer.py:86-92): nearest-neighbour warp of the moving label map by
 * `flow` (B=1,(D,H,W,3) channels-last) and the per-label voxel counts Dice needs.
 * counts (3*(nlabels+1) int64, zeroed here): [0]=n(pred==l), [1]=n(true==l), [2]=n(both==l).
 * warped (D*H*W int16) may be NULL. */
int modet_label_warp_counts(const int16_t* lab_moving, const float* flow, const int16_t* lab_fixed,
                            int16_t* warped, int64_t* counts, int D, int H, int W, int nlabels,
                            modet_stream_t stream);

/* Evaluation tail, part 2 (utils.py:108-150 jacobian_determinant_vxm, infer.py:89-90): counts[b] (int64, zeroed here) =
 * number of voxels of sample b whose Jacobian determinant of (identity + flow) is <= 0.  flow (B,D,H,W,3) channels-last
 * fp32; fp64 arithmetic in the reference's operation order (np.gradient differences, first-row expansion), so the count
 * is integer-exact.  det_out (B,D,H,W fp64) may be NULL.  Every dim must be >= 2 (np.gradient's own requirement). */
int modet_jacdet_nonpos_count(const float* flow, int64_t* counts, double* det_out, int B, int D, int H, int W,
                              modet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 STORAGE / fp32 ACCUMULATE variants of the ConvInsBlock chain (BASELINE.json configs[4]; csrc/conv3d_bf16.hip).
 * Activations marked "bf16" are (B,D,H,W,C) channels-last arrays of 16-bit brain floats; weights, bias, statistics and
 * weight gradients stay fp32.  `*_bf16` int flags: 0 = that tensor is fp32, 1 = bf16.  Convolutions run on the bf16 matrix
 * pipe (v_mfma_f32_16x16x32_bf16) with fp32 accumulators; fp32 inputs are rounded to bf16 (nearest even) when staged.
 *   fwd:      y (bf16) = conv(x (fp32 | bf16), w) + bias; stats (optional, modet_conv3d_bf16_stats_bytes) = InstanceNorm
 *             partial sums taken from the fp32 accumulators, same buffer format as modet_conv3d_fwd_stats
 *             (consumed by modet_instnorm_lrelu_fwd_stats_bf16).  Cout % 8 == 0; Cin % 8 == 0 (bf16 x) or % 4 == 0 (fp32 x).
 *   bwd_data: d_x (fp32 | bf16) from d_y (bf16).
 *   bwd_weight: d_w, d_bias (fp32) from x (fp32 | bf16) and d_y (bf16). */
/* which kernel the bf16 entry points run for this launch shape (pass as modet_conv3d_kernel_family): 1 = tiled
 * (conv3d_bf16_kernel / conv3d_bf16_wgrad_kernel), 2 = z-marching with one bf16 piece (conv_x3_kernel<.., NPC = 1>,
 * conv_x3_wgrad_kernel<.., NPC = 1>): the few-channel full-resolution layers, HBM-bound */
int modet_conv3d_bf16_kernel_family(int B, int D, int H, int W, int Cin, int Cout, int pass, int x_bf16);
size_t modet_conv3d_bf16_ws_bytes(int Cin, int Cout);
size_t modet_conv3d_bf16_stats_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modet_conv3d_bf16_fwd(const void* x, int x_bf16, const float* w, const float* bias, void* y, void* ws, size_t ws_bytes,
                          float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                          modet_stream_t stream, modet_step_ctx_t* step);
int modet_conv3d_bf16_bwd_data(const void* d_y, const float* w, void* d_x, int dx_bf16, void* ws, size_t ws_bytes,
                               int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step);
size_t modet_conv3d_bf16_bwd_weight_ws_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modet_conv3d_bf16_bwd_weight(const void* x, int x_bf16, const void* d_y, float* d_w, float* d_bias, void* ws,
                                 size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream);
/* modet_conv3d_bf16_bwd_weight without its two reduction launches: they are queued and run by
 * modet_conv3d_wgrad_defer_flush together with the fp32 ones (same contract as modet_conv3d_bwd_weight_defer). */
int modet_conv3d_bf16_bwd_weight_defer(const void* x, int x_bf16, const void* d_y, float* d_w, float* d_bias, void* ws,
                                       size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                       modet_step_ctx_t* step);
/* InstanceNorm3d + LeakyReLU(0.1) on a bf16 raw conv output x: forward from the conv's statistics buffer (y fp32 | bf16,
 * mean / rstd (B*C) fp32 out); backward d_x (bf16) from d_y (fp32 | bf16), x, mean, rstd. */
size_t modet_instnorm_bf16_ws_bytes(int B, int64_t V, int C);
int modet_instnorm_lrelu_fwd_stats_bf16(const void* x, void* y, int y_bf16, float* mean, float* rstd, const float* stats,
                                        size_t stats_bytes, int B, int64_t V, int C, float eps, modet_stream_t stream);
/* modet_instnorm_lrelu_fwd_stats_bf16 with a bf16 output AND the AvgPool3d(2) of the output's fp32 values (before the rounding)
 * in the same pass: a level's output block whose features are stored as bf16.  pooled: (B, D/2, H/2, W/2, C) fp32. */
int modet_instnorm_lrelu_fwd_stats_pool_bf16(const void* x, void* y_bf16, float* pooled, float* mean, float* rstd, const float* stats,
                                             size_t stats_bytes, int B, int D, int H, int W, int C, float eps, modet_stream_t stream);
int modet_instnorm_lrelu_bwd_bf16(const void* d_y, int dy_bf16, const void* x, const float* mean, const float* rstd,
                                  void* d_x, void* ws, size_t ws_bytes, int B, int64_t V, int C, modet_stream_t stream);
/* modet_instnorm_lrelu_bwd_pool (above) for the bf16 chain: the gradient of a level's OUTPUT block,
 * unpool(g_pooled) / 8 + [add_a ; add_b] (all fp32), is formed inside the two backward passes; x / d_x are bf16.  C % 8 == 0. */
int modet_instnorm_lrelu_bwd_pool_bf16(const float* g_pooled, const float* add_a, const float* add_b, int Bh, const void* x,
                                       const float* mean, const float* rstd, void* d_x, void* ws, size_t ws_bytes, int B, int D,
                                       int H, int W, int C, modet_stream_t stream);
/* element-wise cast between fp32 and bf16 (round to nearest even), n % 8 == 0 */
int modet_cast_bf16(const void* x, void* y, int64_t n, int to_bf16, modet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PR++ Correlation3D ("Baseline methods/PR++/models.py":205-232; kernel_size 3, d = 3, sw = 1, sf = 2), SURVEY.md 8(f):
 *   corr[b][t][p] = (1/27) sum_c box3(mov)[b,p,c] * box3(fix)[b, p + 2*off(t), c],  t = 9i+3j+k, off = (i-1,j-1,k-1),
 * box3 = zero-padded 3x3x3 box sum (fix's also on the ring just outside the volume, as the reference's padding 3 does).
 * mov, fix (B,D,H,W,C) channels-last, C % 4 == 0; corr / d_corr (B,27,D,H,W). */
size_t modet_corr3d_ws_bytes(int B, int D, int H, int W, int C);
int modet_corr3d_fwd(const float* mov, const float* fix, float* corr, void* ws, size_t ws_bytes, int B, int D, int H,
                     int W, int C, modet_stream_t stream);
int modet_corr3d_bwd(const float* mov, const float* fix, const float* d_corr, float* d_mov, float* d_fix, void* ws,
                     size_t ws_bytes, int B, int D, int H, int W, int C, modet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MODET_HIP_H */
