// ProjectionLayer = Linear(Cin -> dim) + LayerNorm(dim) on channels-last voxels, fused (one pass over HBM).
// reference: ModeT/models.py:230-241 (permute -> nn.Linear -> nn.LayerNorm(eps=1e-5, affine)).
// (Cin, dim) per level 1..5: (8,6) (16,6) (32,12) (64,24) (128,48).
//
// The model's five shapes run on the grouped kernels (proj_ln_*_g_kernel, further down): G lanes per voxel
// (1 at levels 1-2, 4/8/16 at levels 3-5), weights broadcast from LDS, d_W as an MFMA contraction over voxels.
// Any other (Cin % 4 == 0, dim in {6,12,24,48}) takes the generic thread-per-voxel kernels at the top of the file.
//
// Backward recomputes z = Wx+b and the LN statistics (cheaper than saving them), writes d_x, and reduces the
// five parameter gradients in two deterministic stages (per-workgroup partials -> fixed-order fp64 sum).
#include "common.h"

namespace {

constexpr int BLK = 256;
constexpr int MAXW = 48 * 128;

// bf16 storage of the input / output (BASELINE.json configs[4]): X16 / Y16 = the tensor holds bf16 (channels-last, two per 32-bit
// word); the values are widened while loaded / rounded to nearest even while stored, every product and sum in between is fp32
__device__ __forceinline__ float4 ld4x(const float* __restrict__ base, int64_t el, bool x16) {      // 4 consecutive elements from el
  if (x16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + el);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(base + el);
}
__device__ __forceinline__ unsigned short bf16_rne(float v) {
  const __bf16 b = (__bf16)v;
  return __builtin_bit_cast(unsigned short, b);
}

template <int DIM, bool X16 = false>
__device__ __forceinline__ void linear_ln(const float* __restrict__ xbase, int64_t xel, const float* __restrict__ Ws /*[Cin][DIM]*/,
                                          const float* __restrict__ bs, int Cin, float eps, float (&zh)[DIM],
                                          float& rstd) {
  float z[DIM];
#pragma unroll
  for (int o = 0; o < DIM; ++o) z[o] = bs[o];
  for (int c = 0; c < Cin; c += 4) {
    const float4 xv = ld4x(xbase, xel + c, X16);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int o = 0; o < DIM; ++o) z[o] = fmaf(xs[j], Ws[(c + j) * DIM + o], z[o]);
  }
  float mu = 0.f;
#pragma unroll
  for (int o = 0; o < DIM; ++o) mu += z[o];
  mu *= (1.f / DIM);
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < DIM; ++o) { const float d = z[o] - mu; var = fmaf(d, d, var); }
  rstd = rsqrtf(var * (1.f / DIM) + eps);
#pragma unroll
  for (int o = 0; o < DIM; ++o) zh[o] = (z[o] - mu) * rstd;
}

__device__ __forceinline__ void stage_w(float* Ws, float* ps, const float* __restrict__ Wt,
                                        const float* __restrict__ p0, const float* __restrict__ p1,
                                        const float* __restrict__ p2, int Cin, int DIM) {
  for (int i = threadIdx.x; i < Cin * DIM; i += BLK) {
    const int o = i / Cin, c = i - o * Cin;          // Wt is (dim, Cin) row-major
    Ws[c * DIM + o] = Wt[i];
  }
  for (int i = threadIdx.x; i < DIM; i += BLK) {
    ps[i] = p0[i];
    if (p1) ps[DIM + i] = p1[i];
    if (p2) ps[2 * DIM + i] = p2[i];
  }
}

template <int DIM, bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(BLK) void proj_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          int64_t N, int Cin, float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                  // [Cin][DIM]
  float* ps = smem + Cin * DIM;      // bias | gamma | beta
  stage_w(Ws, ps, Wt, bias, gamma, beta, Cin, DIM);
  __syncthreads();
  for (int64_t n = (int64_t)blockIdx.x * BLK + threadIdx.x; n < N; n += (int64_t)gridDim.x * BLK) {
    float zh[DIM], rstd;
    linear_ln<DIM, X16>(x, n * Cin, Ws, ps, Cin, eps, zh, rstd);
    float* yp = y + n * DIM;
#pragma unroll
    for (int o = 0; o < DIM; o += 2) {
      float2 v;
      v.x = fmaf(zh[o], ps[DIM + o], ps[2 * DIM + o]);
      v.y = fmaf(zh[o + 1], ps[DIM + o + 1], ps[2 * DIM + o + 1]);
      if constexpr (Y16) reinterpret_cast<unsigned*>(y)[(n * DIM + o) >> 1] = (unsigned)bf16_rne(v.x) | ((unsigned)bf16_rne(v.y) << 16);
      else *reinterpret_cast<float2*>(yp + o) = v;
    }
  }
}

// Per workgroup: partial[blk][0..DIM) = d_gamma, [DIM..2DIM) = d_beta, [2DIM..3DIM) = d_bias,
// and if REGW: [3DIM .. 3DIM + DIM*CIN) = d_W (row-major (dim,Cin)); otherwise d_z is written to `dz`.
template <int DIM, int CIN_REG>
__global__ __launch_bounds__(BLK) void proj_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ dz_out, float* __restrict__ part,
                                                          int64_t N, int Cin, float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* ps = smem + Cin * DIM;      // bias | gamma
  stage_w(Ws, ps, Wt, bias, gamma, nullptr, Cin, DIM);
  __syncthreads();
  constexpr int NW = CIN_REG > 0 ? DIM * CIN_REG : 1;
  float ag[DIM], ab[DIM], abi[DIM], aw[NW];
#pragma unroll
  for (int o = 0; o < DIM; ++o) { ag[o] = 0.f; ab[o] = 0.f; abi[o] = 0.f; }
#pragma unroll
  for (int i = 0; i < NW; ++i) aw[i] = 0.f;
  for (int64_t n = (int64_t)blockIdx.x * BLK + threadIdx.x; n < N; n += (int64_t)gridDim.x * BLK) {
    float zh[DIM], rstd;
    const float* xp = x + n * Cin;
    linear_ln<DIM>(xp, 0, Ws, ps, Cin, eps, zh, rstd);
    float dzh[DIM];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int o = 0; o < DIM; ++o) {
      const float g = dy[n * DIM + o];
      ag[o] = fmaf(g, zh[o], ag[o]);
      ab[o] += g;
      dzh[o] = g * ps[DIM + o];
      m1 += dzh[o];
      m2 = fmaf(dzh[o], zh[o], m2);
    }
    m1 *= (1.f / DIM); m2 *= (1.f / DIM);
#pragma unroll
    for (int o = 0; o < DIM; ++o) {
      dzh[o] = rstd * (dzh[o] - m1 - zh[o] * m2);        // d loss / d z[o]
      abi[o] += dzh[o];
    }
    if (CIN_REG == 0) {
#pragma unroll
      for (int o = 0; o < DIM; ++o) dz_out[n * DIM + o] = dzh[o];
    }
    if constexpr (CIN_REG > 0) {
#pragma unroll
      for (int c = 0; c < CIN_REG; c += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = 0.f;
#pragma unroll
          for (int o = 0; o < DIM; ++o) {
            s = fmaf(dzh[o], Ws[(c + j) * DIM + o], s);
            aw[o * CIN_REG + c + j] = fmaf(dzh[o], xs[j], aw[o * CIN_REG + c + j]);
          }
          o4[j] = s;
        }
        *reinterpret_cast<float4*>(dx + n * Cin + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
    } else {
      for (int c = 0; c < Cin; c += 4) {
        float o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = 0.f;
#pragma unroll
          for (int o = 0; o < DIM; ++o) s = fmaf(dzh[o], Ws[(c + j) * DIM + o], s);
          o4[j] = s;
        }
        *reinterpret_cast<float4*>(dx + n * Cin + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
    }
  }
  // per-WAVE partials (shuffle reduction only, no barriers): part[(blk*4 + wave)][...]
  const int npart = 3 * DIM + (CIN_REG > 0 ? DIM * CIN_REG : 0);
  const int lane = threadIdx.x & 63;
  float* pp = part + ((int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6)) * npart;
#pragma unroll
  for (int o = 0; o < DIM; ++o) {
    float r = wave_sum(ag[o]);  if (lane == 0) pp[o] = r;
    r = wave_sum(ab[o]);        if (lane == 0) pp[DIM + o] = r;
    r = wave_sum(abi[o]);       if (lane == 0) pp[2 * DIM + o] = r;
  }
  if (CIN_REG > 0) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float r = wave_sum(aw[i]);
      if (lane == 0) pp[3 * DIM + i] = r;
    }
  }
}

// d_W partials for the wide levels: part[chunk][o*Cin + c] = sum_{n in chunk} dz[n][o] * x[n][c]
constexpr int DW_TILE = 32, DW_CHUNK = 64, DW_MAXP = MAXW / BLK;
__global__ __launch_bounds__(BLK) void proj_dw_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                      float* __restrict__ part, int64_t N, int Cin, int DIM) {
  __shared__ float xs[DW_TILE * 128];
  __shared__ float zs[DW_TILE * 48];
  const int P = Cin * DIM;
  float acc[DW_MAXP];
#pragma unroll
  for (int j = 0; j < DW_MAXP; ++j) acc[j] = 0.f;
  const int64_t n0 = (int64_t)blockIdx.x * DW_CHUNK;
  const int64_t n1 = n0 + DW_CHUNK < N ? n0 + DW_CHUNK : N;
  for (int64_t t0 = n0; t0 < n1; t0 += DW_TILE) {
    const int nt = (int)(n1 - t0 < DW_TILE ? n1 - t0 : DW_TILE);
    __syncthreads();
    for (int i = threadIdx.x; i < nt * Cin; i += BLK) xs[i] = x[t0 * Cin + i];
    for (int i = threadIdx.x; i < nt * DIM; i += BLK) zs[i] = dz[t0 * DIM + i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DW_MAXP; ++j) {
      const int p = threadIdx.x + j * BLK;
      if (p < P) {
        const int o = p / Cin, c = p - o * Cin;
        float s = acc[j];
        for (int n = 0; n < nt; ++n) s = fmaf(zs[n * DIM + o], xs[n * Cin + c], s);
        acc[j] = s;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < DW_MAXP; ++j) {
    const int p = threadIdx.x + j * BLK;
    if (p < P) part[(int64_t)blockIdx.x * P + p] = acc[j];
  }
}

// column sums of the partial rows, one wave per output column, fixed assignment + fixed tree, fp64.  Column i of
// `part` (row stride `stride`) goes to the segment it falls in: [0,n0) -> d0, [n0,n0+n1) -> d1, ... (one launch
// finalises d_gamma, d_beta, d_bias and d_W together).
__global__ __launch_bounds__(64) void colsum_kernel(const float* __restrict__ part, int nblk, int stride, float* d0, int n0,
                                                     float* d1, int n1, float* d2, int n2, float* d3, int n3) {
  const int i = blockIdx.x;
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  int b = threadIdx.x;
  for (; b + 192 < nblk; b += 256) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] += (double)part[(int64_t)(b + 64 * u) * stride + i];
  }
  for (; b < nblk; b += 64) s4[0] += (double)part[(int64_t)b * stride + i];
  double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  s = wave_sum_d(s);
  if (threadIdx.x != 0) return;
  if (i < n0) d0[i] = (float)s;
  else if (i < n0 + n1) d1[i - n0] = (float)s;
  else if (i < n0 + n1 + n2) d2[i - n0 - n1] = (float)s;
  else if (i < n0 + n1 + n2 + n3) d3[i - n0 - n1 - n2] = (float)s;
}

// ------------------------------------------------------------------------------------------------ grouped kernels
// The five (Cin, dim) pairs of the model run here.  G lanes share one voxel (G = 1 at levels 1-2 where N is huge,
// 4 / 8 / 16 at levels 3-5 where a thread per voxel would leave the chip empty: level 5 has 1 200 voxels):
//   lane g computes the outputs o = j*G + g (LayerNorm statistics by xor-shuffles inside the group), publishes
//   d_z in LDS, then computes d_x for the channels c = i*G + g.
//   d_W = d_z^T x is an MFMA contraction over the voxels of the tile (A = d_z from LDS, B = the staged x tile), so no
//   lane carries a dim x Cin accumulator matrix; a workgroup accumulates it over all its tiles and writes ONE partial
//   row [d_gamma | d_beta | d_bias | d_W], summed over workgroups by colsum_kernel (fixed order, fp64).
typedef float pf32x4 __attribute__((ext_vector_type(4)));

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int DIM, int CIN, int G>
struct ProjCfg {
  static constexpr int OS = DIM / G, CS = CIN / G;       // outputs / channels per lane
  static constexpr int VPB = BLK / G;                    // voxels per workgroup tile
  static constexpr int DIMS = G == 1 ? DIM : DIM + 1;    // W row stride in LDS (odd: lanes over c hit distinct banks)
  static constexpr int XS = CIN + 4;                     // x row stride in LDS
  static constexpr int OT = (DIM + 15) / 16, DZS = OT * 16 + 4;
  static constexpr int CT = (CIN + 15) / 16;
  static constexpr int PAIRS = OT * CT;                  // 16x16 tiles of d_W
  static constexpr int NSPLIT = PAIRS >= 4 ? 1 : 4 / PAIRS;   // waves splitting the voxels of one tile pair
  static constexpr int NACC = (PAIRS + 3) / 4;
  static constexpr int NPART = 3 * DIM + DIM * CIN;
  static_assert(DIM % G == 0 && CIN % G == 0 && CIN % 4 == 0 && VPB % (4 * NSPLIT) == 0, "shape");
};

template <int DIM, int CIN, int G>
__device__ __forceinline__ void stage_group(float* Ws, float* ps, float* xs, const float* __restrict__ Wt,
                                            const float* __restrict__ p0, const float* __restrict__ p1,
                                            const float* __restrict__ p2) {
  using C = ProjCfg<DIM, CIN, G>;
  for (int i = threadIdx.x; i < CIN * DIM; i += BLK) {
    const int o = i / CIN, c = i - o * CIN;              // Wt is (dim, Cin) row-major
    Ws[c * C::DIMS + o] = Wt[i];
  }
  for (int i = threadIdx.x; i < DIM; i += BLK) {
    ps[i] = p0[i];
    ps[DIM + i] = p1[i];
    if (p2) ps[2 * DIM + i] = p2[i];
  }
  (void)xs;
}

// x tile -> LDS (coalesced float4), rows past N zero filled
template <int DIM, int CIN, int G, bool X16 = false>
__device__ __forceinline__ void load_x_tile(float* xs, const float* __restrict__ x, int64_t n0, int64_t N) {
  using C = ProjCfg<DIM, CIN, G>;
  constexpr int Q = CIN / 4;
  for (int i = threadIdx.x; i < C::VPB * Q; i += BLK) {
    const int v = i / Q, c4 = i - v * Q;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + v < N) val = ld4x(x, (n0 + v) * CIN + c4 * 4, X16);
    *reinterpret_cast<float4*>(xs + v * C::XS + c4 * 4) = val;
  }
}

// z = Wx + b for this lane's outputs, LayerNorm statistics over the group
template <int DIM, int CIN, int G>
__device__ __forceinline__ void linear_ln_group(const float* xrow, const float* Ws, const float* ps, int g, float eps,
                                                float (&zh)[DIM / G], float& rstd) {
  using C = ProjCfg<DIM, CIN, G>;
  float z[C::OS];
#pragma unroll
  for (int j = 0; j < C::OS; ++j) z[j] = ps[j * G + g];
#pragma unroll 2
  for (int c = 0; c < CIN; c += 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xrow + c);
    const float xq[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < C::OS; ++j) z[j] = fmaf(xq[q], Ws[(c + q) * C::DIMS + j * G + g], z[j]);
  }
  float mu = 0.f;
#pragma unroll
  for (int j = 0; j < C::OS; ++j) mu += z[j];
  mu = group_sum<G>(mu) * (1.f / DIM);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < C::OS; ++j) { const float d = z[j] - mu; var = fmaf(d, d, var); }
  var = group_sum<G>(var);
  rstd = rsqrtf(var * (1.f / DIM) + eps);
#pragma unroll
  for (int j = 0; j < C::OS; ++j) zh[j] = (z[j] - mu) * rstd;
}

template <int DIM, int CIN, int G, bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(BLK) void proj_ln_fwd_g_kernel(const float* __restrict__ x_, const float* __restrict__ Wt,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y_,
                                                            int64_t N, float eps, const float* __restrict__ x2_ = nullptr,
                                                            float* __restrict__ y2_ = nullptr) {
  // gridDim.y == 2: the layer applied to two inputs (q and k projections of a level) in one launch -- at levels 3-5 one use is
  // 75-600 workgroups of a latency-bound kernel, two of them side by side cost the time of one
  const float* __restrict__ x = blockIdx.y ? x2_ : x_;
  float* __restrict__ y = blockIdx.y ? y2_ : y_;
  using C = ProjCfg<DIM, CIN, G>;
  __shared__ __attribute__((aligned(16))) float Ws[CIN * C::DIMS];
  __shared__ float ps[3 * DIM];
  __shared__ __attribute__((aligned(16))) float xs[C::VPB * C::XS];
  stage_group<DIM, CIN, G>(Ws, ps, xs, Wt, bias, gamma, beta);
  const int v = threadIdx.x / G, g = threadIdx.x % G;
  const int64_t ntiles = cdiv64(N, (int64_t)C::VPB);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t n0 = t * C::VPB, n = n0 + v;
    __syncthreads();
    load_x_tile<DIM, CIN, G, X16>(xs, x, n0, N);
    __syncthreads();
    float zh[C::OS], rstd;
    linear_ln_group<DIM, CIN, G>(xs + v * C::XS, Ws, ps, g, eps, zh, rstd);
    if (n < N) {
#pragma unroll
      for (int j = 0; j < C::OS; ++j) {
        const int o = j * G + g;
        const float val = fmaf(zh[j], ps[DIM + o], ps[2 * DIM + o]);
        if constexpr (Y16) reinterpret_cast<unsigned short*>(y)[n * DIM + o] = bf16_rne(val);
        else y[n * DIM + o] = val;
      }
    }
  }
}

template <int DIM, int CIN, int G, bool X16 = false>
__global__ __launch_bounds__(BLK) void proj_ln_bwd_g_kernel(const float* __restrict__ x_, const float* __restrict__ Wt,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy_, float* __restrict__ dx_,
                                                            float* __restrict__ part_, int64_t N, float eps,
                                                            const float* __restrict__ x2_ = nullptr,
                                                            const float* __restrict__ dy2_ = nullptr,
                                                            float* __restrict__ dx2_ = nullptr, float* __restrict__ part2_ = nullptr) {
  // gridDim.y == 2: both uses of a paired projection in one launch (see proj_ln_fwd_g_kernel)
  const float* __restrict__ x = blockIdx.y ? x2_ : x_;
  const float* __restrict__ dy = blockIdx.y ? dy2_ : dy_;
  float* __restrict__ dx = blockIdx.y ? dx2_ : dx_;
  float* __restrict__ part = blockIdx.y ? part2_ : part_;
  using C = ProjCfg<DIM, CIN, G>;
  __shared__ __attribute__((aligned(16))) float Ws[CIN * C::DIMS];
  __shared__ float ps[2 * DIM];
  __shared__ __attribute__((aligned(16))) float xs[C::VPB * C::XS + 16];
  __shared__ __attribute__((aligned(16))) float dzs[C::VPB * C::DZS];
  __shared__ float red[4 * (3 * DIM > 256 * C::NACC ? 3 * DIM : 256 * C::NACC)];
  stage_group<DIM, CIN, G>(Ws, ps, xs, Wt, bias, gamma, nullptr);
  for (int i = threadIdx.x; i < C::VPB * C::DZS; i += BLK) dzs[i] = 0.f;      // padding rows of the MFMA A operand
  for (int i = threadIdx.x; i < 16; i += BLK) xs[C::VPB * C::XS + i] = 0.f;
  const int v = threadIdx.x / G, g = threadIdx.x % G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  float ag[C::OS], ab[C::OS], abi[C::OS];
#pragma unroll
  for (int j = 0; j < C::OS; ++j) { ag[j] = 0.f; ab[j] = 0.f; abi[j] = 0.f; }
  pf32x4 acc[C::NACC];
#pragma unroll
  for (int a = 0; a < C::NACC; ++a) acc[a] = (pf32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t ntiles = cdiv64(N, (int64_t)C::VPB);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t n0 = t * C::VPB, n = n0 + v;
    __syncthreads();                                   // previous tile's MFMA reads of xs / dzs are done
    load_x_tile<DIM, CIN, G, X16>(xs, x, n0, N);
    __syncthreads();
    float zh[C::OS], rstd;
    linear_ln_group<DIM, CIN, G>(xs + v * C::XS, Ws, ps, g, eps, zh, rstd);
    float dzh[C::OS];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < C::OS; ++j) {
      const int o = j * G + g;
      const float gy = n < N ? dy[n * DIM + o] : 0.f;
      ag[j] = fmaf(gy, zh[j], ag[j]);
      ab[j] += gy;
      dzh[j] = gy * ps[DIM + o];
      m1 += dzh[j];
      m2 = fmaf(dzh[j], zh[j], m2);
    }
    m1 = group_sum<G>(m1) * (1.f / DIM);
    m2 = group_sum<G>(m2) * (1.f / DIM);
#pragma unroll
    for (int j = 0; j < C::OS; ++j) {
      dzh[j] = rstd * (dzh[j] - m1 - zh[j] * m2);      // d loss / d z[o]
      abi[j] += dzh[j];
      dzs[v * C::DZS + j * G + g] = dzh[j];
    }
    __syncthreads();
    // d_x for the channels c = i*G + g
    if (n < N) {
      if constexpr (G == 1) {
#pragma unroll
        for (int c = 0; c < CIN; c += 4) {
          float o4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < DIM; ++o) sacc = fmaf(dzh[o], Ws[(c + q) * C::DIMS + o], sacc);
            o4[q] = sacc;
          }
          *reinterpret_cast<float4*>(dx + n * CIN + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
      } else {
        float da[C::CS];
#pragma unroll
        for (int i = 0; i < C::CS; ++i) da[i] = 0.f;
#pragma unroll 4
        for (int o = 0; o < DIM; ++o) {
          const float d = dzs[v * C::DZS + o];
#pragma unroll
          for (int i = 0; i < C::CS; ++i) da[i] = fmaf(d, Ws[(i * G + g) * C::DIMS + o], da[i]);
        }
#pragma unroll
        for (int i = 0; i < C::CS; ++i) dx[n * CIN + i * G + g] = da[i];
      }
    }
    // d_W += d_z^T x over the tile's voxels: D[o][c] += A[o][voxel] * B[voxel][c]
    if constexpr (C::NSPLIT == 1) {
#pragma unroll
      for (int a = 0; a < C::NACC; ++a) {
        const int pr = wave + 4 * a;
        if (pr < C::PAIRS) {
          const int ot = pr / C::CT, ct = pr % C::CT;
#pragma unroll 4
          for (int ks = 0; ks < C::VPB / 4; ++ks) {
            const float av = dzs[(ks * 4 + lk) * C::DZS + ot * 16 + li];
            const float bv = xs[(ks * 4 + lk) * C::XS + ct * 16 + li];
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a], 0, 0, 0);
          }
        }
      }
    } else {
      const int pr = wave % C::PAIRS, sub = wave / C::PAIRS;
      const int ot = pr / C::CT, ct = pr % C::CT;
#pragma unroll 4
      for (int ks = sub; ks < C::VPB / 4; ks += C::NSPLIT) {
        const float av = dzs[(ks * 4 + lk) * C::DZS + ot * 16 + li];
        const float bv = xs[(ks * 4 + lk) * C::XS + ct * 16 + li];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[0], 0, 0, 0);
      }
    }
  }

  // ---- one partial row per workgroup: [d_gamma | d_beta | d_bias | d_W (dim, Cin)]
  float* prow = part + (int64_t)blockIdx.x * C::NPART;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < C::OS; ++j) {
    float r0 = ag[j], r1 = ab[j], r2 = abi[j];
#pragma unroll
    for (int o = G; o < 64; o <<= 1) { r0 += __shfl_xor(r0, o, 64); r1 += __shfl_xor(r1, o, 64); r2 += __shfl_xor(r2, o, 64); }
    if (lane < G) {
      const int o = j * G + lane;
      red[wave * 3 * DIM + o] = r0; red[wave * 3 * DIM + DIM + o] = r1; red[wave * 3 * DIM + 2 * DIM + o] = r2;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * DIM; i += BLK)
    prow[i] = ((red[i] + red[3 * DIM + i]) + red[2 * 3 * DIM + i]) + red[3 * 3 * DIM + i];
  if constexpr (C::NSPLIT == 1) {
#pragma unroll
    for (int a = 0; a < C::NACC; ++a) {
      const int pr = wave + 4 * a;
      if (pr < C::PAIRS) {
        const int ot = pr / C::CT, ct = pr % C::CT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int o = ot * 16 + lk * 4 + j, c = ct * 16 + li;
          if (o < DIM && c < CIN) prow[3 * DIM + o * CIN + c] = acc[a][j];
        }
      }
    }
  } else {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave * 256 + (lk * 4 + j) * 16 + li] = acc[0][j];
    __syncthreads();
    for (int i = threadIdx.x; i < C::PAIRS * 256; i += BLK) {
      const int pr = i >> 8, e = i & 255;
      float sacc = red[pr * 256 + e];
#pragma unroll
      for (int k = 1; k < C::NSPLIT; ++k) sacc += red[(pr + k * C::PAIRS) * 256 + e];
      const int ot = pr / C::CT, ct = pr % C::CT;
      const int o = ot * 16 + (e >> 4), c = ct * 16 + (e & 15);
      if (o < DIM && c < CIN) prow[3 * DIM + o * CIN + c] = sacc;
    }
  }
}

inline int group_of(int Cin, int dim) {      // 0: not one of the model's five (Cin, dim) pairs -> generic kernels
  if (dim == 6 && (Cin == 8 || Cin == 16)) return 1;
  if (dim == 12 && Cin == 32) return 4;
  if (dim == 24 && Cin == 64) return 8;
  if (dim == 48 && Cin == 128) return 16;
  return 0;
}
inline int group_grid(int64_t N, int G) {
  int64_t tiles = cdiv64(N, (int64_t)(BLK / G));
  return (int)(tiles < 1024 ? tiles : 1024);
}

inline int bwd_grid(int64_t N) {
  int64_t g = cdiv64(N, (int64_t)BLK);
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}
inline int reg_cin(int, int) { return 0; }     // generic path: d_z to the workspace, d_W by proj_dw_kernel

}  // namespace

namespace {
template <bool X16, bool Y16>
int proj_fwd_launch(const float* x, const float* Wt, const float* bias, const float* gamma, const float* beta, float* y, int64_t N,
                    int Cin, int dim, float eps, hipStream_t s) {
  const size_t sh = ((size_t)Cin * dim + 3 * dim) * sizeof(float);
  const int grid = flat_grid(N, BLK);
  const int G = group_of(Cin, dim);
  if (G > 1) {          // levels 3-5: several lanes per voxel
    const int gg = group_grid(N, G);
    if (G == 4) hipLaunchKernelGGL((proj_ln_fwd_g_kernel<12, 32, 4, X16, Y16>), dim3(gg), dim3(BLK), 0, s, x, Wt, bias, gamma, beta, y, N, eps);
    else if (G == 8) hipLaunchKernelGGL((proj_ln_fwd_g_kernel<24, 64, 8, X16, Y16>), dim3(gg), dim3(BLK), 0, s, x, Wt, bias, gamma, beta, y, N, eps);
    else hipLaunchKernelGGL((proj_ln_fwd_g_kernel<48, 128, 16, X16, Y16>), dim3(gg), dim3(BLK), 0, s, x, Wt, bias, gamma, beta, y, N, eps);
    return modet_launch_status();
  }
  switch (dim) {
    case 6:  hipLaunchKernelGGL((proj_ln_fwd_kernel<6, X16, Y16>),  dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    case 12: hipLaunchKernelGGL((proj_ln_fwd_kernel<12, X16, Y16>), dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    case 24: hipLaunchKernelGGL((proj_ln_fwd_kernel<24, X16, Y16>), dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    case 48: hipLaunchKernelGGL((proj_ln_fwd_kernel<48, X16, Y16>), dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    default: return MODET_ERR_UNSUPPORTED;
  }
  return modet_launch_status();
}
}  // namespace

extern "C" {

int modet_proj_ln_fwd_t(const void* x, int x_bf16, const float* Wt, const float* bias, const float* gamma, const float* beta,
                        void* y, int y_bf16, int64_t N, int Cin, int dim, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(Wt); MODET_CHECK_PTR(bias); MODET_CHECK_PTR(gamma); MODET_CHECK_PTR(beta);
  MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(N > 0 && Cin > 0 && dim > 0);
  if (Cin % 4 != 0 || Cin > 128) return MODET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const float* xf = (const float*)x;
  float* yf = (float*)y;
  if (x_bf16 && y_bf16) return proj_fwd_launch<true, true>(xf, Wt, bias, gamma, beta, yf, N, Cin, dim, eps, s);
  if (x_bf16) return proj_fwd_launch<true, false>(xf, Wt, bias, gamma, beta, yf, N, Cin, dim, eps, s);
  if (y_bf16) return proj_fwd_launch<false, true>(xf, Wt, bias, gamma, beta, yf, N, Cin, dim, eps, s);
  return proj_fwd_launch<false, false>(xf, Wt, bias, gamma, beta, yf, N, Cin, dim, eps, s);
}

int modet_proj_ln_fwd(const float* x, const float* Wt, const float* bias, const float* gamma, const float* beta,
                      float* y, int64_t N, int Cin, int dim, float eps, modet_stream_t stream) {
  return modet_proj_ln_fwd_t(x, 0, Wt, bias, gamma, beta, y, 0, N, Cin, dim, eps, stream);
}

int modet_proj_ln_fwd_pair(const float* x1, const float* x2, const float* Wt, const float* bias, const float* gamma,
                           const float* beta, float* y1, float* y2, int64_t N, int Cin, int dim, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x1); MODET_CHECK_PTR(x2); MODET_CHECK_PTR(Wt); MODET_CHECK_PTR(bias); MODET_CHECK_PTR(gamma); MODET_CHECK_PTR(beta);
  MODET_CHECK_PTR(y1); MODET_CHECK_PTR(y2);
  MODET_CHECK_DIM(N > 0 && Cin > 0 && dim > 0);
  hipStream_t s = (hipStream_t)stream;
  const int G = group_of(Cin, dim);
  if (G > 1) {
    const int gg = group_grid(N, G);
    if (G == 4) hipLaunchKernelGGL((proj_ln_fwd_g_kernel<12, 32, 4, false, false>), dim3(gg, 2), dim3(BLK), 0, s, x1, Wt, bias, gamma, beta, y1, N, eps, x2, y2);
    else if (G == 8) hipLaunchKernelGGL((proj_ln_fwd_g_kernel<24, 64, 8, false, false>), dim3(gg, 2), dim3(BLK), 0, s, x1, Wt, bias, gamma, beta, y1, N, eps, x2, y2);
    else hipLaunchKernelGGL((proj_ln_fwd_g_kernel<48, 128, 16, false, false>), dim3(gg, 2), dim3(BLK), 0, s, x1, Wt, bias, gamma, beta, y1, N, eps, x2, y2);
    return modet_launch_status();
  }
  const int rc = modet_proj_ln_fwd(x1, Wt, bias, gamma, beta, y1, N, Cin, dim, eps, stream);
  return rc != MODET_OK ? rc : modet_proj_ln_fwd(x2, Wt, bias, gamma, beta, y2, N, Cin, dim, eps, stream);
}

size_t modet_proj_ln_bwd_ws_bytes(int64_t N, int Cin, int dim) {
  if (const int G = group_of(Cin, dim)) return (size_t)group_grid(N, G) * (3 * dim + dim * Cin) * sizeof(float);
  const int rc = reg_cin(Cin, dim);
  size_t fl = (size_t)bwd_grid(N) * (BLK / 64) * (3 * dim + (rc ? dim * Cin : 0));
  if (!rc) fl += (size_t)N * dim + (size_t)cdiv64(N, DW_CHUNK) * Cin * dim;
  return fl * sizeof(float);
}

// The layer applied to TWO inputs with the same parameters (ModeT projects the fixed and the moving feature map of a level
// with one ProjectionLayer, models.py:371-372): both data gradients, and the parameter gradients of both uses summed by
// ONE fixed-order fp64 column sum over the partial rows of the two backward launches -- instead of two reductions and
// four element-wise additions of their results.
size_t modet_proj_ln_bwd_pair_ws_bytes(int64_t N, int Cin, int dim) {
  if (!group_of(Cin, dim)) return 0;                  // only the grouped kernels (every level of the model)
  return 2 * modet_proj_ln_bwd_ws_bytes(N, Cin, dim);
}

int modet_proj_ln_bwd_pair(const float* x1, const float* d_y1, float* d_x1, const float* x2, const float* d_y2, float* d_x2,
                           const float* Wt, const float* bias, const float* gamma, float* d_Wt, float* d_bias,
                           float* d_gamma, float* d_beta, void* ws, size_t ws_bytes, int64_t N, int Cin, int dim, float eps,
                           modet_stream_t stream) {
  return modet_proj_ln_bwd_pair_t(x1, 0, d_y1, d_x1, x2, 0, d_y2, d_x2, Wt, bias, gamma, d_Wt, d_bias, d_gamma, d_beta, ws, ws_bytes, N,
                                  Cin, dim, eps, stream);
}

int modet_proj_ln_bwd_pair_t(const void* x1v, int x1_bf16, const float* d_y1, float* d_x1, const void* x2v, int x2_bf16,
                             const float* d_y2, float* d_x2, const float* Wt, const float* bias, const float* gamma, float* d_Wt,
                             float* d_bias, float* d_gamma, float* d_beta, void* ws, size_t ws_bytes, int64_t N, int Cin, int dim,
                             float eps, modet_stream_t stream) {
  const float* x1 = (const float*)x1v;
  const float* x2 = (const float*)x2v;
  MODET_CHECK_PTR(x1); MODET_CHECK_PTR(d_y1); MODET_CHECK_PTR(d_x1); MODET_CHECK_PTR(x2); MODET_CHECK_PTR(d_y2);
  MODET_CHECK_PTR(d_x2); MODET_CHECK_PTR(Wt); MODET_CHECK_PTR(bias); MODET_CHECK_PTR(gamma); MODET_CHECK_PTR(ws);
  const bool reduce_here = d_Wt || d_bias || d_gamma || d_beta;       // all four NULL: the partial rows stay in ws (header)
  if (reduce_here) { MODET_CHECK_PTR(d_Wt); MODET_CHECK_PTR(d_bias); MODET_CHECK_PTR(d_gamma); MODET_CHECK_PTR(d_beta); }
  MODET_CHECK_DIM(N > 0 && Cin > 0 && dim > 0);
  const int G = group_of(Cin, dim);
  if (Cin % 4 != 0 || Cin > 128 || !G) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_proj_ln_bwd_pair_ws_bytes(N, Cin, dim)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int gg = group_grid(N, G), row = 3 * dim + dim * Cin;
  float* gpart = (float*)ws;
  if (G > 1 && (x1_bf16 != 0) == (x2_bf16 != 0)) {      // levels 3-5: the two uses side by side in ONE launch (grid.y = 2)
    float* part2 = gpart + (size_t)gg * row;
#define LAUNCH_G2(D_, C_, G_)                                                                                                     \
    do {                                                                                                                          \
      if (x1_bf16) hipLaunchKernelGGL((proj_ln_bwd_g_kernel<D_, C_, G_, true>), dim3(gg, 2), dim3(BLK), 0, s, x1, Wt, bias, gamma,  \
                                      d_y1, d_x1, gpart, N, eps, x2, d_y2, d_x2, part2);                                          \
      else hipLaunchKernelGGL((proj_ln_bwd_g_kernel<D_, C_, G_, false>), dim3(gg, 2), dim3(BLK), 0, s, x1, Wt, bias, gamma, d_y1,   \
                              d_x1, gpart, N, eps, x2, d_y2, d_x2, part2);                                                        \
    } while (0)
    if (G == 4) LAUNCH_G2(12, 32, 4);
    else if (G == 8) LAUNCH_G2(24, 64, 8);
    else LAUNCH_G2(48, 128, 16);
#undef LAUNCH_G2
  } else
  for (int u = 0; u < 2; ++u) {
    const float* x = u ? x2 : x1;
    const float* d_y = u ? d_y2 : d_y1;
    float* d_x = u ? d_x2 : d_x1;
    float* part = gpart + (size_t)u * gg * row;
    const bool x16 = (u ? x2_bf16 : x1_bf16) != 0;
#define LAUNCH_G(D_, C_, G_)                                                                                                      \
    do {                                                                                                                          \
      if (x16) hipLaunchKernelGGL((proj_ln_bwd_g_kernel<D_, C_, G_, true>), dim3(gg), dim3(BLK), 0, s, x, Wt, bias, gamma, d_y, d_x, \
                                  part, N, eps);                                                                                 \
      else hipLaunchKernelGGL((proj_ln_bwd_g_kernel<D_, C_, G_, false>), dim3(gg), dim3(BLK), 0, s, x, Wt, bias, gamma, d_y, d_x,  \
                              part, N, eps);                                                                                     \
    } while (0)
    if (G == 1 && Cin == 8) LAUNCH_G(6, 8, 1);
    else if (G == 1) LAUNCH_G(6, 16, 1);
    else if (G == 4) LAUNCH_G(12, 32, 4);
    else if (G == 8) LAUNCH_G(24, 64, 8);
    else LAUNCH_G(48, 128, 16);
#undef LAUNCH_G
  }
  if (reduce_here)
    hipLaunchKernelGGL(colsum_kernel, dim3(row), dim3(64), 0, s, (const float*)gpart, 2 * gg, row, d_gamma, dim, d_beta, dim,
                       d_bias, dim, d_Wt, dim * Cin);
  return modet_launch_status();
}

int64_t modet_proj_ln_bwd_pair_partial_rows(int64_t N, int Cin, int dim) {
  const int G = group_of(Cin, dim);
  return (N > 0 && G) ? 2 * (int64_t)group_grid(N, G) : 0;
}

// ---- every leaf reduction of a backward pass in ONE launch.  The attention's d_rpb (two launches per level) and the paired
// projection's d_gamma / d_beta / d_bias / d_W (one per level) are column sums of per-workgroup partial rows whose only consumer
// is the gradient buffer: nothing downstream waits for them, but on one stream -- and in the captured graph -- each is a 5-8 us
// node of its own, fifteen per step.  A caller that passes NULL for those outputs keeps the partial rows in its workspace and
// hands the jobs over here at the end of the pass: one wave per output column, fixed lane assignment and a fixed tree in fp64
// (deterministic), as colsum_kernel.
struct LeafTable { modet_leaf_job_t j[MODET_LEAF_MAX_JOBS]; int first[MODET_LEAF_MAX_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(64) void leaf_reduce_many_kernel(const LeafTable t) {
  const int c = blockIdx.x;
  int ji = 0;
  while (ji + 1 < t.njobs && c >= t.first[ji + 1]) ++ji;
  const modet_leaf_job_t& J = t.j[ji];
  const int i = c - t.first[ji];
  const float* p = J.part + (int64_t)(i / J.col_group) * J.col_group_stride + (i % J.col_group);
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t o = 0; o < J.outer; ++o) {
    const float* po = p + o * J.outer_stride;
    int64_t b = threadIdx.x;
    for (; b + 192 < J.rows; b += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += (double)po[(b + 64 * u) * J.row_stride];
    }
    for (; b < J.rows; b += 64) s4[0] += (double)po[b * J.row_stride];
  }
  double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  s = wave_sum_d(s);
  if (threadIdx.x != 0) return;
  int k = i;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (k < J.n[u]) { J.dst[u][k] = (float)s; return; }
    k -= J.n[u];
  }
}

int modet_leaf_reduce_many(const modet_leaf_job_t* jobs, int njobs, modet_stream_t stream) {
  if (njobs == 0) return MODET_OK;
  MODET_CHECK_PTR(jobs);
  MODET_CHECK_DIM(njobs > 0);
  hipStream_t s = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += MODET_LEAF_MAX_JOBS) {
    LeafTable t;
    t.njobs = njobs - j0 < MODET_LEAF_MAX_JOBS ? njobs - j0 : MODET_LEAF_MAX_JOBS;
    int total = 0;
    for (int j = 0; j < t.njobs; ++j) {
      const modet_leaf_job_t& J = jobs[j0 + j];
      MODET_CHECK_PTR(J.part);
      MODET_CHECK_DIM(J.outer > 0 && J.rows > 0 && J.ncols > 0 && J.col_group > 0);
      int nsum = 0;
      for (int u = 0; u < 4; ++u) {
        MODET_CHECK_DIM(J.n[u] >= 0);
        if (J.n[u] > 0) MODET_CHECK_PTR(J.dst[u]);
        nsum += J.n[u];
      }
      MODET_CHECK_DIM(nsum == J.ncols);
      t.j[j] = J;
      t.first[j] = total;
      total += J.ncols;
    }
    t.first[t.njobs] = total;
    hipLaunchKernelGGL(leaf_reduce_many_kernel, dim3(total), dim3(64), 0, s, t);
  }
  return modet_launch_status();
}

int modet_proj_ln_bwd(const float* x, const float* Wt, const float* bias, const float* gamma, const float* d_y,
                      float* d_x, float* d_Wt, float* d_bias, float* d_gamma, float* d_beta, void* ws, size_t ws_bytes,
                      int64_t N, int Cin, int dim, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(Wt); MODET_CHECK_PTR(bias); MODET_CHECK_PTR(gamma); MODET_CHECK_PTR(d_y);
  MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(d_Wt); MODET_CHECK_PTR(d_bias); MODET_CHECK_PTR(d_gamma); MODET_CHECK_PTR(d_beta);
  MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(N > 0 && Cin > 0 && dim > 0);
  if (Cin % 4 != 0 || Cin > 128) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_proj_ln_bwd_ws_bytes(N, Cin, dim)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (const int G = group_of(Cin, dim)) {
    const int gg = group_grid(N, G);
    float* gpart = (float*)ws;
#define LAUNCH_G(D_, C_, G_) hipLaunchKernelGGL((proj_ln_bwd_g_kernel<D_, C_, G_>), dim3(gg), dim3(BLK), 0, s, x, Wt, bias, gamma, \
                                                d_y, d_x, gpart, N, eps)
    if (G == 1 && Cin == 8) LAUNCH_G(6, 8, 1);
    else if (G == 1) LAUNCH_G(6, 16, 1);
    else if (G == 4) LAUNCH_G(12, 32, 4);
    else if (G == 8) LAUNCH_G(24, 64, 8);
    else LAUNCH_G(48, 128, 16);
#undef LAUNCH_G
    hipLaunchKernelGGL(colsum_kernel, dim3(3 * dim + dim * Cin), dim3(64), 0, s, (const float*)gpart, gg, 3 * dim + dim * Cin,
                       d_gamma, dim, d_beta, dim, d_bias, dim, d_Wt, dim * Cin);
    return modet_launch_status();
  }
  const int grid = bwd_grid(N);
  const int rc = reg_cin(Cin, dim);
  const int npart = 3 * dim + (rc ? dim * Cin : 0);
  float* part = (float*)ws;
  const int nwp = grid * (BLK / 64);              // per-wave partial rows
  float* dz = part + (size_t)nwp * npart;
  float* dwpart = dz + (size_t)N * dim;
  const size_t sh = ((size_t)Cin * dim + 2 * dim) * sizeof(float);
#define LAUNCH_BWD(D_, R_) hipLaunchKernelGGL((proj_ln_bwd_kernel<D_, R_>), dim3(grid), dim3(BLK), sh, s, x, Wt, bias, \
                                              gamma, d_y, d_x, dz, part, N, Cin, eps)
  if (dim == 6) LAUNCH_BWD(6, 0);
  else if (dim == 12) LAUNCH_BWD(12, 0);
  else if (dim == 24) LAUNCH_BWD(24, 0);
  else if (dim == 48) LAUNCH_BWD(48, 0);
  else return MODET_ERR_UNSUPPORTED;
#undef LAUNCH_BWD
  const bool regw = rc != 0;
  if (regw) {
    hipLaunchKernelGGL(colsum_kernel, dim3(3 * dim + dim * Cin), dim3(64), 0, s, (const float*)part, nwp, npart, d_gamma, dim,
                       d_beta, dim, d_bias, dim, d_Wt, dim * Cin);
  } else {
    hipLaunchKernelGGL(colsum_kernel, dim3(3 * dim), dim3(64), 0, s, (const float*)part, nwp, npart, d_gamma, dim, d_beta, dim,
                       d_bias, dim, (float*)nullptr, 0);
    const int nchunk = (int)cdiv64(N, DW_CHUNK);
    hipLaunchKernelGGL(proj_dw_kernel, dim3(nchunk), dim3(BLK), 0, s, x, dz, dwpart, N, Cin, dim);
    hipLaunchKernelGGL(colsum_kernel, dim3(dim * Cin), dim3(64), 0, s, (const float*)dwpart, nchunk, dim * Cin, d_Wt, dim * Cin,
                       (float*)nullptr, 0, (float*)nullptr, 0, (float*)nullptr, 0);
  }
  return modet_launch_status();
}

}  // extern "C"
