// NCC_vxm(win=9) and Grad3d('l2') forward + gradient (reference ModeT/losses.py:34-94, :6-31).
//
// The reference evaluates five dense 9x9x9 all-ones conv3d (729 taps each, ~36 GFLOP at 160x192x160).  The box filter is
// separable; here each direction is ONE z-marching kernel (ncc_march_kernel): a workgroup owns a 24 x 32 (y, x) tile,
// every thread keeps the last WIN planes of its 5 halo voxels in REGISTERS (the z box sum of the five products is
// register arithmetic, each input plane is loaded once per chunk), then the W and H box sums run over two small LDS
// arrays and the pointwise tail (cc + the backward coefficients, resp. the final gradient formula) finishes the plane:
// forward reads I, J and writes three coefficient volumes, backward reads those + I, J and writes d_J -- no five-volume
// intermediates (round 2: four passes, 31 N floats of traffic, 0.21 ms; now 2 launches + the scalar finalize).
// The backward uses that the zero-padded box sum S is self-adjoint:
//   d(-mean cc)/dJ = g * ( S(cB) + 2 J S(cD) + I S(cE) ),  g = -1/N,
//   cE = d cc/d IJ_sum, cD = d cc/d J2_sum, cB = d cc/d J_sum  (pointwise in the five sums).
// Scalar losses are reduced in two deterministic stages (workgroup partials -> fixed-order fp64 sum).
#include "common.h"

namespace {

constexpr int BLK = 256;
constexpr int WIN = 9, PAD = 4;
constexpr float WINSZ = 729.f;

struct Dims { int B, D, H, W; };

// 32-bit form: a 64-bit division by a run-time value is a ~100-instruction software routine (three of them made
// grad3d_kernel VALU-bound: 14.7 M elements x ~350 instructions)
__device__ __forceinline__ void decode32(unsigned i, const Dims d, int& z, int& y, int& x) {
  const unsigned t = i / (unsigned)d.W;
  x = (int)(i - t * (unsigned)d.W);
  const unsigned u = t / (unsigned)d.H;
  y = (int)(t - u * (unsigned)d.H);
  z = (int)(u % (unsigned)d.D);
}
__device__ __forceinline__ void decode(int64_t i, const Dims d, int& z, int& y, int& x) {
  x = (int)(i % d.W);
  const int64_t t = i / d.W;
  y = (int)(t % d.H);
  z = (int)((t / d.H) % d.D);
}

// ------------------------------------------------------------------------------------------------ z-marching form
// FWD: in0 = I, in1 = J (NIN = 2); the five window sums {I, J, I^2, J^2, I J} -> cc (loss partial per workgroup) and the
//      coefficient volumes coef = {cB, cD, cE}.   BWD: in0 = coef (three stacked volumes, NIN = 3) -> S(cB), S(cD), S(cE)
//      -> d_J = g (S(cB) + 2 J S(cD) + I S(cE)).   Each window sum is the plain WIN^3-term sum (z, then x, then y), no
//      running differences.  Zero padding = predicated loads.
constexpr int MT_Y = 24, MT_X = 32;
#ifndef NCC_VARIANT
#define NCC_VARIANT 0          // tuning builds (tools/variants.sh): 1 no W / H passes, 2 no loads after the prologue, 4 no z sums
#endif
// NOUT consecutive W_-term window sums of v[0 .. NOUT + W_ - 2]: the values common to all windows are added once (purely
// additive: another association of the same terms, no running differences).  T = float or a float pair (v_pk_add_f32).
typedef float f2 __attribute__((ext_vector_type(2)));
template <int W_, int NOUT, typename T>
__device__ __forceinline__ void window_sums(const T (&v)[NOUT + W_ - 1], T (&o)[NOUT]) {
  if constexpr (W_ >= NOUT) {
    T core = v[NOUT - 1];
#pragma unroll
    for (int k = NOUT; k < W_; ++k) core += v[k];
#pragma unroll
    for (int m = 0; m < NOUT; ++m) {
      T a = core;
#pragma unroll
      for (int k = m; k < NOUT - 1; ++k) a += v[k];
#pragma unroll
      for (int k = W_; k < W_ + m; ++k) a += v[k];
      o[m] = a;
    }
  } else {                                       // windows shorter than the run: no common core
#pragma unroll
    for (int m = 0; m < NOUT; ++m) {
      T a = v[m];
#pragma unroll
      for (int k = 1; k < W_; ++k) a += v[m + k];
      o[m] = a;
    }
  }
}

// The kernels are VALU-issue bound (measured with compile-time variants: the loads are a tenth of the time), so the
// quantities travel in PAIRS through registers and LDS and every sum is a packed fp32 instruction (v_pk_add_f32 /
// v_pk_fma_f32: two lanes' worth of work per issue):  FWD: (I, J) -> pairs (sum I, sum J), (sum I^2, sum J^2) + the single
// sum I J;  BWD: pair (cB, cD) + the single cE.
template <int W_, bool FWD>
__global__ __launch_bounds__(BLK, 2) void ncc_march_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                           const float* __restrict__ Ivol, const float* __restrict__ Jvol,
                                                           float* __restrict__ out, float* __restrict__ part, Dims d,
                                                           int64_t N, int tiles_x, int tiles_y, int nchunk, int zc,
                                                           float g) {
  constexpr int P = W_ / 2, HY = MT_Y + 2 * P, HXW = MT_X + 2 * P, NVOX = HY * HXW;
  constexpr int NIN = FWD ? 2 : 3, NPQ = FWD ? 2 : 1;           // inputs; pairs of summed quantities (+ one single)
  constexpr int NV = (NVOX + BLK - 1) / BLK;                     // halo voxels per thread
  constexpr int RPT = MT_Y / (BLK / MT_X);                       // output rows per thread in the H pass (3)
  constexpr float WINSZ_ = (float)(W_ * W_ * W_);
  __shared__ __attribute__((aligned(16))) f2 zsP[NPQ][HY * HXW + 2];                // + a dummy slot
  __shared__ __attribute__((aligned(16))) float zsS[HY * HXW + 4];
  __shared__ __attribute__((aligned(16))) f2 twP[NPQ][HY * MT_X];
  __shared__ __attribute__((aligned(16))) float twS[HY * MT_X];
  __shared__ float red[BLK / 64];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int x0 = (t % tiles_x) * MT_X; t /= tiles_x;
  const int y0 = (t % tiles_y) * MT_Y; t /= tiles_y;
  const int zs0 = (t % nchunk) * zc;
  const int b = t / nchunk;
  const int ze = zs0 + zc < d.D ? zs0 + zc : d.D;
  const int64_t plane = (int64_t)d.H * d.W;
  const float* src[NIN];
  src[0] = in0 + (int64_t)b * d.D * plane;
  if constexpr (FWD) src[1] = in1 + (int64_t)b * d.D * plane;
  else { src[1] = in0 + N + (int64_t)b * d.D * plane; src[2] = in0 + 2 * N + (int64_t)b * d.D * plane; }

  // this thread's halo voxels: byte offset inside a plane (out of range for the zero padding) and LDS slot (a dummy
  // slot past the tile for the few threads beyond the halo tile's end): every load is a buffer load through a
  // descriptor of ONE plane -- out-of-range offsets and planes outside the volume (num_records = 0) read 0 -- so a
  // plane's loads are issued back to back, with no branch and no select
  constexpr unsigned OOB = 0x80000000u;
  unsigned goff[NV];
  int lidx[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = tid + j * BLK;
    const int hy = i / HXW, hx = i - hy * HXW;
    const int y = y0 + hy - P, x = x0 + hx - P;
    const bool ok = i < NVOX && y >= 0 && y < d.H && x >= 0 && x < d.W;
    goff[j] = ok ? (unsigned)(y * d.W + x) * 4u : OOB;
    lidx[j] = i < NVOX ? i : NVOX;
  }
  const unsigned plane_bytes = (unsigned)plane * 4u;
  // the last W_ planes of this thread's halo voxels: inputs 0 and 1 as a pair, input 2 (BWD) on its own
  f2 ringP[NV][W_], nxtP[NV];
  float ringS[FWD ? 1 : NV][W_], nxtS[FWD ? 1 : NV];
  auto load_plane = [&](int z, f2 (&dp)[NV], float (&ds)[FWD ? 1 : NV]) {
    const bool zin = z >= 0 && z < d.D;
#pragma unroll
    for (int v = 0; v < NIN; ++v) {
      const uint64_t a = reinterpret_cast<uint64_t>(src[v] + (int64_t)(zin ? z : 0) * plane);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                  (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)),
          0, (int)(zin ? plane_bytes : 0u), 0x00020000);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float val = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)goff[j], 0, 0));
        if (v == 0) dp[j].x = val;
        else if (v == 1) dp[j].y = val;
        else ds[FWD ? 0 : j] = val;
      }
    }
  };
  // The window sums do not care which ring slot is the oldest plane, only the REPLACEMENT does: slot (z - zs0) % W_ is
  // overwritten through a wave-uniform switch (a handful of register moves), nothing is shifted.
  // prologue: planes zs0 - P .. zs0 + P - 1 into ring slots 1 .. W_-1 (slot 0 is filled by the first iteration)
#pragma unroll
  for (int k = 1; k < W_; ++k) {
    f2 tp[NV];
    float ts[FWD ? 1 : NV];
    load_plane(zs0 - P + k - 1, tp, ts);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      ringP[j][k] = tp[j];
      if constexpr (!FWD) ringS[j][k] = ts[j];
    }
  }
  load_plane(zs0 + P, nxtP, nxtS);

  float lsum = 0.f;
  int slot = 0;
  for (int z = zs0; z < ze; ++z) {
    // ring <- planes z - P .. z + P: the newest one replaces the oldest
#pragma unroll
    for (int k = 0; k < W_; ++k) {
      if (slot == k) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          ringP[j][k] = nxtP[j];
          if constexpr (!FWD) ringS[j][k] = nxtS[j];
        }
      }
    }
    slot = slot + 1 == W_ ? 0 : slot + 1;
#if !(NCC_VARIANT & 2)
    if (z + 1 < ze) load_plane(z + 1 + P, nxtP, nxtS);           // in flight during this plane's arithmetic
#endif
    // ---- z box sums of this thread's halo voxels -> LDS
#pragma unroll
    for (int j = 0; j < ((NCC_VARIANT & 4) ? 1 : NV); ++j) {
      f2 ab = {0.f, 0.f};
      if constexpr (FWD) {
        f2 ce = {0.f, 0.f};
        float f = 0.f;
#pragma unroll
        for (int k = 0; k < W_; ++k) {
          const f2 ij = ringP[j][k];
          ab += ij;                                              // (sum I, sum J)
          ce = __builtin_elementwise_fma(ij, ij, ce);            // (sum I^2, sum J^2)
          f = fmaf(ij.x, ij.y, f);                               // sum I J
        }
        zsP[0][lidx[j]] = ab; zsP[NPQ - 1][lidx[j]] = ce; zsS[lidx[j]] = f;
      } else {
        float c = 0.f;
#pragma unroll
        for (int k = 0; k < W_; ++k) { ab += ringP[j][k]; c += ringS[j][k]; }
        zsP[0][lidx[j]] = ab; zsS[lidx[j]] = c;
      }
    }
    __syncthreads();
    // ---- W pass: (halo row r, 4 consecutive outputs) from a register window of 4 + W_ - 1 values
#if !(NCC_VARIANT & 1)
    {
      const int r = tid >> 3, c0 = (tid & 7) * 4;
      if (r < HY) {
#pragma unroll
        for (int q = 0; q < NPQ; ++q) {
          f2 v[4 + W_ - 1], o[4];
#pragma unroll
          for (int k = 0; k < 4 + W_ - 1; ++k) v[k] = zsP[q][r * HXW + c0 + k];
          window_sums<W_, 4>(v, o);
#pragma unroll
          for (int m = 0; m < 4; ++m) twP[q][r * MT_X + c0 + m] = o[m];
        }
        float v[4 + W_ - 1], o[4];
#pragma unroll
        for (int k = 0; k < 4 + W_ - 1; ++k) v[k] = zsS[r * HXW + c0 + k];
        window_sums<W_, 4>(v, o);
        *reinterpret_cast<float4*>(&twS[r * MT_X + c0]) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
#endif
    __syncthreads();
    // ---- H pass + pointwise tail: column x, RPT consecutive output rows
    {
      const int x = tid & (MT_X - 1), r0 = (tid / MT_X) * RPT;
      f2 sp[NPQ][RPT];
      float ss[RPT];
#pragma unroll
      for (int q = 0; q < NPQ; ++q) {
        f2 v[RPT + W_ - 1];
#pragma unroll
        for (int k = 0; k < RPT + W_ - 1; ++k) v[k] = twP[q][(r0 + k) * MT_X + x];
        window_sums<W_, RPT>(v, sp[q]);
      }
      {
        float v[RPT + W_ - 1];
#pragma unroll
        for (int k = 0; k < RPT + W_ - 1; ++k) v[k] = twS[(r0 + k) * MT_X + x];
        window_sums<W_, RPT>(v, ss);
      }
#pragma unroll
      for (int m = 0; m < RPT; ++m) {
        const int y = y0 + r0 + m, xg = x0 + x;
        if (y >= d.H || xg >= d.W) continue;
        const int64_t o = ((int64_t)b * d.D + z) * plane + (int64_t)y * d.W + xg;
        if constexpr (FWD) {
          // the reference's own (expanded) order, losses.py:85-91
          const float I_sum = sp[0][m].x, J_sum = sp[0][m].y, I2_sum = sp[NPQ - 1][m].x, J2_sum = sp[NPQ - 1][m].y, IJ_sum = ss[m];
          // (six IEEE divisions per voxel were a third of this kernel's instructions: the window size divides as a
          // multiplication by its rounded reciprocal, 1 / den is one v_rcp_f32 (1 ulp) shared by cc, cE and cD)
          constexpr float RW = 1.f / WINSZ_;
          const float u_I = I_sum * RW, u_J = J_sum * RW;
          const float cross = IJ_sum - u_J * I_sum - u_I * J_sum + u_I * u_J * WINSZ_;
          const float I_var = I2_sum - 2.f * u_I * I_sum + u_I * u_I * WINSZ_;
          const float J_var = J2_sum - 2.f * u_J * J_sum + u_J * u_J * WINSZ_;
          const float rden = __builtin_amdgcn_rcpf(I_var * J_var + 1e-5f);
          const float cc = cross * cross * rden;
          lsum += cc;
          if (out) {
            const float cE = 2.f * cross * rden;
            const float cD = -cc * I_var * rden;
            const float cB = -(cE * I_sum + 2.f * cD * J_sum) * RW;
            out[o] = cB; out[N + o] = cD; out[2 * N + o] = cE;
          }
        } else {
          out[o] = g * (sp[0][m].x + 2.f * Jvol[o] * sp[0][m].y + Ivol[o] * ss[m]);
        }
      }
    }
  }
  if constexpr (FWD) {
    const float r = block_sum(lsum, red);
    if (tid == 0) part[blockIdx.x] = r;
  }
}

struct MarchPlan { int tiles_x, tiles_y, nchunk, zc, grid; };
inline MarchPlan march_plan(int B, int D, int H, int W, int win) {
  MarchPlan p;
  p.tiles_x = cdiv(W, MT_X); p.tiles_y = cdiv(H, MT_Y);
  // z chunks: every chunk re-reads win - 1 planes, so as long as the chip allows (>= ~3 workgroups per CU)
  const int cols = B * p.tiles_x * p.tiles_y;
  int n = cdiv(768, cols);
  if (n < 1) n = 1;
  int zc = cdiv(D, n);
  if (zc < win) zc = win < D ? win : D;
  p.zc = zc; p.nchunk = cdiv(D, zc);
  p.grid = cols * p.nchunk;
  return p;
}


// loss[0] = scale * sum(part[0..n))  (fp64, fixed order); optionally adds into loss (accumulate != 0)
__global__ void scalar_finalize_kernel(const float* __restrict__ part, int n, double scale, float* __restrict__ loss) {
  __shared__ double sm[BLK];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += BLK) s += (double)part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0;
    for (int i = 0; i < BLK; ++i) r += sm[i];
    loss[0] = (float)(r * scale);
  }
}

// ------------------------------------------------------------------------------------------------ NCC, any window
// NCC_vxm(win=[wz, wy, wx]) for the windows the z-march kernel is not instantiated for -- even, anisotropic or > 9 voxels.
// The reference pads EVERY axis by p = floor(win[0] / 2) (losses.py:57), so the five window sums live on a grid of
// (D + 2p - wz + 1) x (H + 2p - wy + 1) x (W + 2p - wx + 1) voxels that differs from the volume's unless the window is cubic
// and odd; cc is averaged over THAT grid.  Plain separable form through a workspace (this path is about the API contract, not
// speed): products -> box sums along x, y, z (window [o - p, o - p + w) clipped to the axis = zero padding) -> pointwise cc +
// coefficients -> the adjoint box sums z, y, x (voxel i collects the outputs o in [i + p - w + 1, i + p]) -> d_J.
struct AxisBox { int nvol, outer, n_src, n_dst, inner, w, off; };     // dst[j] = sum_{t = j + off}^{j + off + w - 1} src[t], t clipped
__global__ __launch_bounds__(BLK) void ncc_box_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, const AxisBox a,
                                                           int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLK) {
    const int in = (int)(i % a.inner);
    const int64_t t1 = i / a.inner;
    const int j = (int)(t1 % a.n_dst);
    const int64_t vo = t1 / a.n_dst;                        // (volume, outer) index, shared by src and dst
    const int lo = max(j + a.off, 0), hi = min(j + a.off + a.w, a.n_src);
    const float* sp = src + (vo * a.n_src) * a.inner + in;
    float acc = 0.f;
    for (int t = lo; t < hi; ++t) acc += sp[(int64_t)t * a.inner];
    dst[i] = acc;
  }
}
__global__ __launch_bounds__(BLK) void ncc_products_kernel(const float* __restrict__ I, const float* __restrict__ J,
                                                           float* __restrict__ v5, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLK) {
    const float a = I[i], b = J[i];
    v5[i] = a; v5[N + i] = b; v5[2 * N + i] = a * a; v5[3 * N + i] = b * b; v5[4 * N + i] = a * b;
  }
}
// the reference's own (expanded) order, losses.py:81-91, with true divisions; coef = {cB, cD, cE} as in ncc_march_kernel
__global__ __launch_bounds__(BLK) void ncc_cc_kernel(const float* __restrict__ s5, float* __restrict__ coef, float* __restrict__ part,
                                                     int64_t M, float win_size) {
  __shared__ float red[16];
  float lsum = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < M; i += (int64_t)gridDim.x * BLK) {
    const float I_sum = s5[i], J_sum = s5[M + i], I2_sum = s5[2 * M + i], J2_sum = s5[3 * M + i], IJ_sum = s5[4 * M + i];
    const float u_I = I_sum / win_size, u_J = J_sum / win_size;
    const float cross = IJ_sum - u_J * I_sum - u_I * J_sum + u_I * u_J * win_size;
    const float I_var = I2_sum - 2.f * u_I * I_sum + u_I * u_I * win_size;
    const float J_var = J2_sum - 2.f * u_J * J_sum + u_J * u_J * win_size;
    const float den = I_var * J_var + 1e-5f;
    const float cc = cross * cross / den;
    lsum += cc;
    if (coef) {
      const float cE = 2.f * cross / den;
      const float cD = -cc * I_var / den;
      const float cB = -(cE * I_sum + 2.f * cD * J_sum) / win_size;
      coef[i] = cB; coef[M + i] = cD; coef[2 * M + i] = cE;
    }
  }
  const float r = block_sum(lsum, red);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ __launch_bounds__(BLK) void ncc_dj_kernel(const float* __restrict__ g3, const float* __restrict__ I,
                                                     const float* __restrict__ J, float* __restrict__ dJ, int64_t N, float g) {
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLK)
    dJ[i] = g * (g3[i] + 2.f * J[i] * g3[N + i] + I[i] * g3[2 * N + i]);
}
struct BoxGeo { int p, Do, Ho, Wo; int64_t M, cap; };
inline bool box_geo(int B, int D, int H, int W, int wz, int wy, int wx, BoxGeo& g) {
  if (wz < 1 || wy < 1 || wx < 1) return false;
  g.p = wz / 2;
  g.Do = D + 2 * g.p - wz + 1; g.Ho = H + 2 * g.p - wy + 1; g.Wo = W + 2 * g.p - wx + 1;
  if (g.Do < 1 || g.Ho < 1 || g.Wo < 1) return false;
  g.M = (int64_t)B * g.Do * g.Ho * g.Wo;
  g.cap = (int64_t)B * (D > g.Do ? D : g.Do) * (H > g.Ho ? H : g.Ho) * (W > g.Wo ? W : g.Wo);      // any intermediate volume fits
  return true;
}
constexpr int NCC_BOX_PARTS = 1024;

// Grad3d: flow (B,3,D,H,W) planar (CS = 1: d = {3 B, D, H, W}) or channels-last (B,D,H,W,3) (CS = 3: d = {B, D, H, W}, the
// three components of a voxel are neighbours in memory and the spatial neighbours CS elements apart -- the layout the
// model's last composition writes, so a training step needs neither the planar copy of the flow nor the copy of its
// gradient back).  'l2': loss = (mean dH^2 + mean dD^2 + mean dW^2)/3; 'l1' (L1 = true): the same with |d| instead of d^2
// (losses.py:11-27); the gradient of |t| at t = 0 is 0, as torch.abs's backward (sign(0) = 0).  df = gscale * d loss / d f
// (gscale = the loss term's weight, train.py:127-129; the product with 1.0f is exact).
template <bool L1, int CS>
__global__ __launch_bounds__(BLK) void grad3d_kernel(const float* __restrict__ f, float* __restrict__ df,
                                                     float* __restrict__ part, Dims d, int64_t N, float inD, float inH,
                                                     float inW, float gscale) {
  __shared__ float red[BLK / 64];
  const int64_t sD = (int64_t)d.H * d.W * CS, sH = (int64_t)d.W * CS;
  float lsum = 0.f;
  const bool small = N < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLK) {
    int z, y, x;
    if (small) decode32((unsigned)i / (unsigned)CS, d, z, y, x);
    else decode(i / CS, d, z, y, x);
    const float v = f[i];
    float g = 0.f;
    // pen(t) = t^2 (l2) or |t| (l1); dpen(t) = t (the factor 2 is applied at the end) or sign(t)
    auto pen = [](float t) { return L1 ? fabsf(t) : t * t; };
    auto dpen = [](float t) { return L1 ? (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)) : t; };
    // the six neighbours are loaded unconditionally and back to back: a missing neighbour re-reads the voxel itself, so its
    // difference is an exact 0 and adds nothing (a load per `if (inside)` was a memory round trip each, in series)
    const float zp = f[z + 1 < d.D ? i + sD : i], zm = f[z > 0 ? i - sD : i];
    const float yp = f[y + 1 < d.H ? i + sH : i], ym = f[y > 0 ? i - sH : i];
    const float xp = f[x + 1 < d.W ? i + CS : i], xm = f[x > 0 ? i - CS : i];
    { const float t = zp - v; lsum = fmaf(pen(t), inD, lsum); g -= dpen(t) * inD; }
    { const float t = v - zm; g += dpen(t) * inD; }
    { const float t = yp - v; lsum = fmaf(pen(t), inH, lsum); g -= dpen(t) * inH; }
    { const float t = v - ym; g += dpen(t) * inH; }
    { const float t = xp - v; lsum = fmaf(pen(t), inW, lsum); g -= dpen(t) * inW; }
    { const float t = v - xm; g += dpen(t) * inW; }
    if (df) df[i] = (g * (L1 ? 1.f / 3.f : 2.f / 3.f)) * gscale;
  }
  const float r = block_sum(lsum, red);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

inline int red_grid(int64_t N) {
  int g = flat_grid(N, BLK);
  return g > 2048 ? 2048 : g;
}

}  // namespace

extern "C" {

size_t modet_ncc_ws_bytes(int B, int D, int H, int W) {
  const int64_t N = (int64_t)B * D * H * W;
  // the three coefficient volumes of the backward + one loss partial per workgroup of the forward kernel
  return ((size_t)3 * N + (size_t)march_plan(B, D, H, W, 3).grid + 64) * sizeof(float);
}

int modet_ncc_fwd_bwd_win_scaled(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes, int B,
                                 int D, int H, int W, int win, float grad_scale, modet_stream_t stream) {
  MODET_CHECK_PTR(I); MODET_CHECK_PTR(J); MODET_CHECK_PTR(loss); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0);
  if (win != 3 && win != 5 && win != 7 && win != 9) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_ncc_ws_bytes(B, D, H, W)) return MODET_ERR_WORKSPACE;
  if ((int64_t)D * H * W >= (1ll << 31)) return MODET_ERR_DIM;
  if ((int64_t)H * W * 4 >= (1ll << 31)) return MODET_ERR_DIM;      // 32-bit byte offsets inside a plane (buffer loads, OOB sentinel)
  hipStream_t s = (hipStream_t)stream;
  const Dims d{B, D, H, W};
  const int64_t N = (int64_t)B * D * H * W;
  float* coef = (float*)ws;
  float* part = coef + 3 * N;
  const MarchPlan p = march_plan(B, D, H, W, win);
#define NCC_GO(W_)                                                                                                          \
  do {                                                                                                                     \
    hipLaunchKernelGGL((ncc_march_kernel<W_, true>), dim3(p.grid), dim3(BLK), 0, s, I, J, (const float*)nullptr,          \
                       (const float*)nullptr, d_J ? coef : (float*)nullptr, part, d, N, p.tiles_x, p.tiles_y, p.nchunk, p.zc, 0.f); \
    hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(BLK), 0, s, (const float*)part, p.grid, -1.0 / (double)N, loss); \
    if (d_J)                                                                                                               \
      hipLaunchKernelGGL((ncc_march_kernel<W_, false>), dim3(p.grid), dim3(BLK), 0, s, (const float*)coef,                \
                         (const float*)nullptr, I, J, d_J, (float*)nullptr, d, N, p.tiles_x, p.tiles_y, p.nchunk, p.zc,    \
                         -grad_scale / (float)N);                                                                          \
  } while (0)
  if (win == 9) NCC_GO(9);
  else if (win == 7) NCC_GO(7);
  else if (win == 5) NCC_GO(5);
  else NCC_GO(3);
#undef NCC_GO
  return modet_launch_status();
}

int modet_ncc_fwd_bwd_win(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes, int B, int D,
                          int H, int W, int win, modet_stream_t stream) {
  return modet_ncc_fwd_bwd_win_scaled(I, J, loss, d_J, ws, ws_bytes, B, D, H, W, win, 1.f, stream);
}

int modet_ncc_fwd_bwd(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes, int B, int D,
                      int H, int W, modet_stream_t stream) {
  return modet_ncc_fwd_bwd_win(I, J, loss, d_J, ws, ws_bytes, B, D, H, W, 9, stream);
}

size_t modet_ncc_box_ws_bytes(int B, int D, int H, int W, int wz, int wy, int wx) {
  BoxGeo g;
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || !box_geo(B, D, H, W, wz, wy, wx, g)) return 0;
  return ((size_t)10 * g.cap + NCC_BOX_PARTS + 64) * sizeof(float);       // two ping-pong buffers of five volumes + loss partials
}

int modet_ncc_fwd_bwd_box(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes, int B, int D,
                          int H, int W, int wz, int wy, int wx, modet_stream_t stream) {
  MODET_CHECK_PTR(I); MODET_CHECK_PTR(J); MODET_CHECK_PTR(loss); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0);
  BoxGeo g;
  if (!box_geo(B, D, H, W, wz, wy, wx, g)) return MODET_ERR_DIM;           // (the reference's conv3d fails on an empty output too)
  if (ws_bytes < modet_ncc_box_ws_bytes(B, D, H, W, wz, wy, wx)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t N = (int64_t)B * D * H * W, M = g.M;
  float* X = (float*)ws;
  float* Y = X + 5 * g.cap;
  float* part = Y + 5 * g.cap;
  auto box = [&](const float* src, float* dst, int nvol, int outer, int n_src, int n_dst, int inner, int w, int off) {
    const AxisBox a{nvol, outer, n_src, n_dst, inner, w, off};
    const int64_t total = (int64_t)nvol * outer * n_dst * inner;
    hipLaunchKernelGGL(ncc_box_axis_kernel, dim3(flat_grid(total, BLK)), dim3(BLK), 0, s, src, dst, a, total);
  };
  hipLaunchKernelGGL(ncc_products_kernel, dim3(flat_grid(N, BLK)), dim3(BLK), 0, s, I, J, X, N);
  // (volume, outer) is one flat index: the five volumes are stacked, so nvol * outer = 5 * B * (dims in front of the axis)
  box(X, Y, 5, B * D * H, W, g.Wo, 1, wx, -g.p);                            // x: (B, D, H, W)   -> (B, D, H, Wo)
  box(Y, X, 5, B * D, H, g.Ho, g.Wo, wy, -g.p);                             // y: (B, D, H, Wo)  -> (B, D, Ho, Wo)
  box(X, Y, 5, B, D, g.Do, g.Ho * g.Wo, wz, -g.p);                          // z: (B, D, Ho, Wo) -> (B, Do, Ho, Wo)
  const float win_size = (float)wz * (float)wy * (float)wx;
  const int gp = flat_grid(M, BLK) < NCC_BOX_PARTS ? flat_grid(M, BLK) : NCC_BOX_PARTS;
  hipLaunchKernelGGL(ncc_cc_kernel, dim3(gp), dim3(BLK), 0, s, (const float*)Y, d_J ? X : (float*)nullptr, part, M, win_size);
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(BLK), 0, s, (const float*)part, gp, -1.0 / (double)M, loss);
  if (d_J) {
    box(X, Y, 3, B, g.Do, D, g.Ho * g.Wo, wz, g.p - wz + 1);                // adjoint z: (B, Do, Ho, Wo) -> (B, D, Ho, Wo)
    box(Y, X, 3, B * D, g.Ho, H, g.Wo, wy, g.p - wy + 1);                   // adjoint y
    box(X, Y, 3, B * D * H, g.Wo, W, 1, wx, g.p - wx + 1);                  // adjoint x: -> (B, D, H, W)
    hipLaunchKernelGGL(ncc_dj_kernel, dim3(flat_grid(N, BLK)), dim3(BLK), 0, s, (const float*)Y, I, J, d_J, N, -1.f / (float)M);
  }
  return modet_launch_status();
}

size_t modet_grad3d_ws_bytes(int, int, int, int) { return 2048 * sizeof(float); }

static int grad3d_launch(const float* flow, float* loss, float* d_flow, void* ws, size_t ws_bytes, int B, int D, int H,
                         int W, int penalty, bool channels_last, float grad_scale, modet_stream_t stream) {
  MODET_CHECK_DIM(penalty == 1 || penalty == 2);
  MODET_CHECK_PTR(flow); MODET_CHECK_PTR(loss); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1);
  if (ws_bytes < modet_grad3d_ws_bytes(B, D, H, W)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const Dims d{channels_last ? B : B * 3, D, H, W};
  const int64_t N = (int64_t)B * 3 * D * H * W;
  const float inD = (float)(1.0 / ((double)B * 3 * (D - 1) * H * W));
  const float inH = (float)(1.0 / ((double)B * 3 * D * (H - 1) * W));
  const float inW = (float)(1.0 / ((double)B * 3 * D * H * (W - 1)));
  const int rg = red_grid(N);
#define GRAD3D_GO(L1_, CS_)                                                                                                 \
  hipLaunchKernelGGL((grad3d_kernel<L1_, CS_>), dim3(rg), dim3(BLK), 0, s, flow, d_flow, (float*)ws, d, N, inD, inH, inW,  \
                     grad_scale)
  if (penalty == 1) { if (channels_last) GRAD3D_GO(true, 3); else GRAD3D_GO(true, 1); }
  else { if (channels_last) GRAD3D_GO(false, 3); else GRAD3D_GO(false, 1); }
#undef GRAD3D_GO
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(BLK), 0, s, (const float*)ws, rg, 1.0 / 3.0, loss);
  return modet_launch_status();
}

int modet_grad3d_fwd_bwd(const float* flow, float* loss, float* d_flow, void* ws, size_t ws_bytes, int B, int D, int H,
                         int W, int penalty, modet_stream_t stream) {
  return grad3d_launch(flow, loss, d_flow, ws, ws_bytes, B, D, H, W, penalty, false, 1.f, stream);
}

int modet_grad3d_fwd_bwd_cl(const float* flow_cl, float* loss, float* d_flow_cl, void* ws, size_t ws_bytes, int B, int D,
                            int H, int W, int penalty, float grad_scale, modet_stream_t stream) {
  return grad3d_launch(flow_cl, loss, d_flow_cl, ws, ws_bytes, B, D, H, W, penalty, true, grad_scale, stream);
}

}  // extern "C"
