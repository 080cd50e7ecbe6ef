// NCC_vxm(win=9) and Grad3d('l2') forward + gradient (reference ModeT/losses.py:34-94, :6-31).
//
// The reference evaluates five dense 9x9x9 all-ones conv3d (729 taps each, ~36 GFLOP at 160x192x160); the box
// filter is separable, so here each is a fused W+H pass over LDS tiles and a D pass with register windows over
// (B,D,H,W) volumes: HBM-bound streams.  The backward uses that the zero-padded box sum S is self-adjoint:
//   d(-mean cc)/dJ = g * ( S(cB) + 2 J S(cD) + I S(cE) ),  g = -1/N,
//   cE = d cc/d IJ_sum, cD = d cc/d J2_sum, cB = d cc/d J_sum  (pointwise in the five sums).
// Scalar losses are reduced in two deterministic stages (workgroup partials -> fixed-order fp64 sum).
#include "common.h"

namespace {

constexpr int BLK = 256;
constexpr int WIN = 9, PAD = 4;
constexpr float WINSZ = 729.f;

struct Dims { int B, D, H, W; };

__device__ __forceinline__ void decode(int64_t i, const Dims d, int& z, int& y, int& x) {
  x = (int)(i % d.W);
  const int64_t t = i / d.W;
  y = (int)(t % d.H);
  z = (int)((t / d.H) % d.D);
}

// W and H box passes in one kernel over 32x32 tiles of a z-slice staged in LDS (+4 halo each side, zeros outside the
// volume): rows are box-summed along W into a second LDS array, then columns along H.  PROD: the inputs are I and J and
// the five box-summed quantities {I, J, I*I, J*J, I*J} are formed on the fly (forward); otherwise NARR stacked volumes
// are filtered as they are (the three coefficient volumes of the backward).  Replaces two full HBM passes
// (a W pass and an H pass) by one.
constexpr int HW_T = 32, HW_H = HW_T + 2 * PAD, HW_LD = HW_H + 1;
template <int NARR, bool PROD>
__global__ __launch_bounds__(BLK) void box_hw_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                     float* __restrict__ out, Dims d, int64_t N, int tiles_w) {
  constexpr int NIN = PROD ? 2 : NARR;
  __shared__ float raw[NIN][HW_H * HW_LD];
  __shared__ float tmp[NARR][HW_H * HW_T];
  const int x0 = (blockIdx.x % tiles_w) * HW_T, y0 = (blockIdx.x / tiles_w) * HW_T;
  const int z = blockIdx.y, b = blockIdx.z;
  const int64_t slice = ((int64_t)b * d.D + z) * d.H * d.W;
  for (int i = threadIdx.x; i < HW_H * HW_H; i += BLK) {
    const int r = i / HW_H, c = i - r * HW_H;
    const int y = y0 + r - PAD, x = x0 + c - PAD;
    const bool ok = y >= 0 && y < d.H && x >= 0 && x < d.W;
    const int64_t o = slice + (int64_t)y * d.W + x;
    if (PROD) {
      raw[0][r * HW_LD + c] = ok ? in0[o] : 0.f;
      raw[1][r * HW_LD + c] = ok ? in1[o] : 0.f;
    } else {
#pragma unroll
      for (int v = 0; v < NARR; ++v) raw[v][r * HW_LD + c] = ok ? in0[(int64_t)v * N + o] : 0.f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW_H * HW_T; i += BLK) {          // W pass: (halo row r, output column c)
    const int r = i / HW_T, c = i - r * HW_T;
    if (PROD) {
      float a = 0.f, bb = 0.f, cc = 0.f, e = 0.f, f = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; ++k) {
        const float iv = raw[0][r * HW_LD + c + k], jv = raw[1][r * HW_LD + c + k];
        a += iv; bb += jv; cc = fmaf(iv, iv, cc); e = fmaf(jv, jv, e); f = fmaf(iv, jv, f);
      }
      tmp[0][i] = a; tmp[1][i] = bb; tmp[2][i] = cc; tmp[3][i] = e; tmp[4][i] = f;
    } else {
#pragma unroll
      for (int v = 0; v < NARR; ++v) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) a += raw[v][r * HW_LD + c + k];
        tmp[v][i] = a;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW_T * HW_T; i += BLK) {          // H pass + store
    const int r = i / HW_T, c = i - r * HW_T;
    const int y = y0 + r, x = x0 + c;
    if (y >= d.H || x >= d.W) continue;
    const int64_t o = slice + (int64_t)y * d.W + x;
#pragma unroll
    for (int v = 0; v < NARR; ++v) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; ++k) a += tmp[v][(r + k) * HW_T + c];
      out[(int64_t)v * N + o] = a;
    }
  }
}


// 9-tap pass along D (axis 0) or H (axis 1) over `nvol` stacked volumes: each thread produces SEG consecutive
// outputs along the axis from a (SEG+8)-value register window (lanes run along W, so every load is coalesced):
// 24 loads per 16 outputs instead of 144, each output still the plain 9-term sum.
constexpr int SEG = 16;
template <int AXIS>
__device__ __forceinline__ void seg_decode(int64_t t, const Dims d, int& z, int& y, int& x, int& b) {
  x = (int)(t % d.W); t /= d.W;
  if (AXIS == 0) {
    y = (int)(t % d.H); t /= d.H;
    const int nseg = (d.D + SEG - 1) / SEG;
    z = (int)(t % nseg) * SEG; b = (int)(t / nseg);
  } else {
    const int nseg = (d.H + SEG - 1) / SEG;
    y = (int)(t % nseg) * SEG; t /= nseg;
    z = (int)(t % d.D); b = (int)(t / d.D);
  }
}
template <int AXIS>
__device__ __forceinline__ void seg_window(const float* __restrict__ vol, const Dims d, int b, int z, int y, int x,
                                           float (&win)[SEG + 2 * PAD]) {
  const int len = AXIS == 0 ? d.D : d.H, p0 = AXIS == 0 ? z : y;
  const int64_t stride = AXIS == 0 ? (int64_t)d.H * d.W : d.W;
  const float* base = vol + (((int64_t)b * d.D + (AXIS == 0 ? 0 : z)) * d.H + (AXIS == 0 ? y : 0)) * d.W + x;
#pragma unroll
  for (int i = 0; i < SEG + 2 * PAD; ++i) {
    const int p = p0 + i - PAD;
    win[i] = (p >= 0 && p < len) ? base[(int64_t)p * stride] : 0.f;
  }
}
__device__ __forceinline__ float win_sum(const float (&win)[SEG + 2 * PAD], int i) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < WIN; ++k) s += win[i + k];
  return s;
}


// last forward pass (along D) fused with cc and the three backward coefficients; arithmetic in the
// reference's own (expanded) order, losses.py:85-91
__global__ __launch_bounds__(BLK) void ncc_pass_d_fwd_kernel(const float* __restrict__ T, float* __restrict__ coef,
                                                             float* __restrict__ part, Dims d, int64_t N,
                                                             int64_t nthreads) {
  __shared__ float red[BLK / 64];
  const int64_t stride = (int64_t)d.H * d.W;
  float lsum = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < nthreads; t += (int64_t)gridDim.x * BLK) {
    int z, y, x, b;
    seg_decode<0>(t, d, z, y, x, b);
    const int64_t o0 = (((int64_t)b * d.D + z) * d.H + y) * d.W + x;
    float s[5][SEG];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
      float win[SEG + 2 * PAD];
      seg_window<0>(T + (int64_t)v * N, d, b, z, y, x, win);
#pragma unroll
      for (int i = 0; i < SEG; ++i) s[v][i] = win_sum(win, i);
    }
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      if (z + i >= d.D) continue;
      const float I_sum = s[0][i], J_sum = s[1][i], I2_sum = s[2][i], J2_sum = s[3][i], IJ_sum = s[4][i];
      const float u_I = I_sum / WINSZ, u_J = J_sum / WINSZ;
      const float cross = IJ_sum - u_J * I_sum - u_I * J_sum + u_I * u_J * WINSZ;
      const float I_var = I2_sum - 2.f * u_I * I_sum + u_I * u_I * WINSZ;
      const float J_var = J2_sum - 2.f * u_J * J_sum + u_J * u_J * WINSZ;
      const float den = I_var * J_var + 1e-5f;
      const float cc = cross * cross / den;
      lsum += cc;
      if (coef) {
        const float cE = 2.f * cross / den;
        const float cD = -cc * I_var / den;
        const float cB = -(cE * I_sum + 2.f * cD * J_sum) / WINSZ;
        const int64_t o = o0 + (int64_t)i * stride;
        coef[o] = cB; coef[N + o] = cD; coef[2 * N + o] = cE;
      }
    }
  }
  const float r = block_sum(lsum, red);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// last backward pass (along D) fused with the final combine
__global__ __launch_bounds__(BLK) void ncc_pass_d_bwd_kernel(const float* __restrict__ T, const float* __restrict__ I,
                                                             const float* __restrict__ J, float* __restrict__ dJ,
                                                             Dims d, int64_t N, float g, int64_t nthreads) {
  const int64_t stride = (int64_t)d.H * d.W;
  for (int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x; t < nthreads; t += (int64_t)gridDim.x * BLK) {
    int z, y, x, b;
    seg_decode<0>(t, d, z, y, x, b);
    const int64_t o0 = (((int64_t)b * d.D + z) * d.H + y) * d.W + x;
    float s[3][SEG];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      float win[SEG + 2 * PAD];
      seg_window<0>(T + (int64_t)v * N, d, b, z, y, x, win);
#pragma unroll
      for (int i = 0; i < SEG; ++i) s[v][i] = win_sum(win, i);
    }
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      if (z + i >= d.D) continue;
      const int64_t o = o0 + (int64_t)i * stride;
      dJ[o] = g * (s[0][i] + 2.f * J[o] * s[1][i] + I[o] * s[2][i]);
    }
  }
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

// Grad3d: flow (B,3,D,H,W) planar.  'l2': loss = (mean dH^2 + mean dD^2 + mean dW^2)/3; 'l1' (L1 = true): the same with
// |d| instead of d^2 (losses.py:11-27); the gradient of |t| at t = 0 is 0, as torch.abs's backward (sign(0) = 0).
template <bool L1>
__global__ __launch_bounds__(BLK) void grad3d_kernel(const float* __restrict__ f, float* __restrict__ df,
                                                     float* __restrict__ part, Dims d, int64_t N, float inD, float inH,
                                                     float inW) {
  __shared__ float red[BLK / 64];
  const int64_t sD = (int64_t)d.H * d.W, sH = d.W;
  float lsum = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < N; i += (int64_t)gridDim.x * BLK) {
    int z, y, x;
    decode(i, d, z, y, x);
    const float v = f[i];
    float g = 0.f;
    // pen(t) = t^2 (l2) or |t| (l1); dpen(t) = t (the factor 2 is applied at the end) or sign(t)
    auto pen = [](float t) { return L1 ? fabsf(t) : t * t; };
    auto dpen = [](float t) { return L1 ? (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)) : t; };
    if (z + 1 < d.D) { const float t = f[i + sD] - v; lsum = fmaf(pen(t), inD, lsum); g -= dpen(t) * inD; }
    if (z > 0)       { const float t = v - f[i - sD]; g += dpen(t) * inD; }
    if (y + 1 < d.H) { const float t = f[i + sH] - v; lsum = fmaf(pen(t), inH, lsum); g -= dpen(t) * inH; }
    if (y > 0)       { const float t = v - f[i - sH]; g += dpen(t) * inH; }
    if (x + 1 < d.W) { const float t = f[i + 1] - v;  lsum = fmaf(pen(t), inW, lsum); g -= dpen(t) * inW; }
    if (x > 0)       { const float t = v - f[i - 1];  g += dpen(t) * inW; }
    if (df) df[i] = g * (L1 ? 1.f / 3.f : 2.f / 3.f);
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
  return ((size_t)10 * N + 2048) * sizeof(float);
}

int modet_ncc_fwd_bwd(const float* I, const float* J, float* loss, float* d_J, void* ws, size_t ws_bytes, int B, int D,
                      int H, int W, modet_stream_t stream) {
  MODET_CHECK_PTR(I); MODET_CHECK_PTR(J); MODET_CHECK_PTR(loss); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0);
  if (ws_bytes < modet_ncc_ws_bytes(B, D, H, W)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const Dims d{B, D, H, W};
  const int64_t N = (int64_t)B * D * H * W;
  float* T1 = (float*)ws;
  float* T2 = T1 + 5 * N;
  float* part = T2 + 5 * N;
  const int64_t nth_d = (int64_t)B * cdiv(D, SEG) * H * W;
  int rg = flat_grid(nth_d, BLK);
  if (rg > 2048) rg = 2048;
  // forward: W+H box sums of the five products (one kernel) -> D pass fused with cc and the loss partials
  const int tw = cdiv(W, HW_T), th = cdiv(H, HW_T);
  const dim3 hwgrid(tw * th, D, B);
  hipLaunchKernelGGL((box_hw_kernel<5, true>), hwgrid, dim3(BLK), 0, s, I, J, T2, d, N, tw);
  hipLaunchKernelGGL(ncc_pass_d_fwd_kernel, dim3(rg), dim3(BLK), 0, s, (const float*)T2, d_J ? T1 : nullptr, part, d, N, nth_d);
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(BLK), 0, s, (const float*)part, rg, -1.0 / (double)N, loss);
  if (d_J) {
    // backward: the box sum is self-adjoint: W+H on the three coefficient volumes, D pass fused with the final formula
    hipLaunchKernelGGL((box_hw_kernel<3, false>), hwgrid, dim3(BLK), 0, s, (const float*)T1, (const float*)nullptr, T2, d, N,
                       tw);
    hipLaunchKernelGGL(ncc_pass_d_bwd_kernel, dim3(flat_grid(nth_d, BLK)), dim3(BLK), 0, s, (const float*)T2, I, J, d_J, d, N,
                       -1.f / (float)N, nth_d);
  }
  return modet_launch_status();
}

size_t modet_grad3d_ws_bytes(int, int, int, int) { return 2048 * sizeof(float); }

int modet_grad3d_fwd_bwd(const float* flow, float* loss, float* d_flow, void* ws, size_t ws_bytes, int B, int D, int H,
                         int W, int penalty, modet_stream_t stream) {
  MODET_CHECK_DIM(penalty == 1 || penalty == 2);
  MODET_CHECK_PTR(flow); MODET_CHECK_PTR(loss); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1);
  if (ws_bytes < modet_grad3d_ws_bytes(B, D, H, W)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const Dims d{B * 3, D, H, W};
  const int64_t N = (int64_t)B * 3 * D * H * W;
  const float inD = (float)(1.0 / ((double)B * 3 * (D - 1) * H * W));
  const float inH = (float)(1.0 / ((double)B * 3 * D * (H - 1) * W));
  const float inW = (float)(1.0 / ((double)B * 3 * D * H * (W - 1)));
  const int rg = red_grid(N);
  if (penalty == 1) hipLaunchKernelGGL(grad3d_kernel<true>, dim3(rg), dim3(BLK), 0, s, flow, d_flow, (float*)ws, d, N, inD, inH, inW);
  else hipLaunchKernelGGL(grad3d_kernel<false>, dim3(rg), dim3(BLK), 0, s, flow, d_flow, (float*)ws, d, N, inD, inH, inW);
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(BLK), 0, s, (const float*)ws, rg, 1.0 / 3.0, loss);
  return modet_launch_status();
}

}  // extern "C"
