// ProjectionLayer = Linear(Cin -> dim) + LayerNorm(dim) on channels-last voxels, fused (one pass over HBM).
// reference: ModeT/models.py:230-241 (permute -> nn.Linear -> nn.LayerNorm(eps=1e-5, affine)).
// (Cin, dim) per level 1..5: (8,6) (16,6) (32,12) (64,24) (128,48).  HBM-bound: one thread per voxel, the
// weight matrix is broadcast from LDS, all `dim` outputs and the LayerNorm statistics stay in registers.
//
// Backward recomputes z = Wx+b and the LN statistics (cheaper than saving them), writes d_x, and reduces the
// five parameter gradients in two deterministic stages (per-workgroup partials -> fixed-order fp64 sum).
#include "common.h"

namespace {

constexpr int BLK = 256;
constexpr int MAXW = 48 * 128;

template <int DIM>
__device__ __forceinline__ void linear_ln(const float* __restrict__ xp, const float* __restrict__ Ws /*[Cin][DIM]*/,
                                          const float* __restrict__ bs, int Cin, float eps, float (&zh)[DIM],
                                          float& rstd) {
  float z[DIM];
#pragma unroll
  for (int o = 0; o < DIM; ++o) z[o] = bs[o];
  for (int c = 0; c < Cin; c += 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xp + c);
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

template <int DIM>
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
    linear_ln<DIM>(x + n * Cin, Ws, ps, Cin, eps, zh, rstd);
    float* yp = y + n * DIM;
#pragma unroll
    for (int o = 0; o < DIM; o += 2) {
      float2 v;
      v.x = fmaf(zh[o], ps[DIM + o], ps[2 * DIM + o]);
      v.y = fmaf(zh[o + 1], ps[DIM + o + 1], ps[2 * DIM + o + 1]);
      *reinterpret_cast<float2*>(yp + o) = v;
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
    linear_ln<DIM>(xp, Ws, ps, Cin, eps, zh, rstd);
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

inline int bwd_grid(int64_t N) {
  int64_t g = cdiv64(N, (int64_t)BLK);
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}
inline int reg_cin(int Cin, int dim) { return (dim == 6 && (Cin == 8 || Cin == 16)) ? Cin : 0; }

}  // namespace

extern "C" {

int modet_proj_ln_fwd(const float* x, const float* Wt, const float* bias, const float* gamma, const float* beta,
                      float* y, int64_t N, int Cin, int dim, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(Wt); MODET_CHECK_PTR(bias); MODET_CHECK_PTR(gamma); MODET_CHECK_PTR(beta);
  MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(N > 0 && Cin > 0 && dim > 0);
  if (Cin % 4 != 0 || Cin > 128) return MODET_ERR_UNSUPPORTED;
  const size_t sh = ((size_t)Cin * dim + 3 * dim) * sizeof(float);
  const int grid = flat_grid(N, BLK);
  hipStream_t s = (hipStream_t)stream;
  switch (dim) {
    case 6:  hipLaunchKernelGGL(proj_ln_fwd_kernel<6>,  dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    case 12: hipLaunchKernelGGL(proj_ln_fwd_kernel<12>, dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    case 24: hipLaunchKernelGGL(proj_ln_fwd_kernel<24>, dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    case 48: hipLaunchKernelGGL(proj_ln_fwd_kernel<48>, dim3(grid), dim3(BLK), sh, s, x, Wt, bias, gamma, beta, y, N, Cin, eps); break;
    default: return MODET_ERR_UNSUPPORTED;
  }
  return modet_launch_status();
}

size_t modet_proj_ln_bwd_ws_bytes(int64_t N, int Cin, int dim) {
  const int rc = reg_cin(Cin, dim);
  size_t fl = (size_t)bwd_grid(N) * (BLK / 64) * (3 * dim + (rc ? dim * Cin : 0));
  if (!rc) fl += (size_t)N * dim + (size_t)cdiv64(N, DW_CHUNK) * Cin * dim;
  return fl * sizeof(float);
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
  if (dim == 6 && Cin == 8) LAUNCH_BWD(6, 8);
  else if (dim == 6 && Cin == 16) LAUNCH_BWD(6, 16);
  else if (dim == 6) LAUNCH_BWD(6, 0);
  else if (dim == 12) LAUNCH_BWD(12, 0);
  else if (dim == 24) LAUNCH_BWD(24, 0);
  else if (dim == 48) LAUNCH_BWD(48, 0);
  else return MODET_ERR_UNSUPPORTED;
#undef LAUNCH_BWD
  const bool regw = (dim == 6 && (Cin == 8 || Cin == 16));
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
