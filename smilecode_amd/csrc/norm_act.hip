// InstanceNorm3d + LeakyReLU (ConvInsBlock, reference ModeT/models.py:135-151), AvgPool3d(2) (:201-219),
// LeakyReLU backward, flat Adam-amsgrad (ModeT/train.py:101), device-scalar scaling.  Channels-last streams.
//
// InstanceNorm statistics are reduced in two deterministic stages: per-workgroup fp32 partial sums over a
// chunk of voxels -> workspace, then one fixed-order fp64 sum per (b,c).  No atomics, so ranks stay bit-consistent.
#include "common.h"

namespace {

constexpr int BLK = 256;
// voxels per partial-sum workgroup: 32 passes of the 256/(C/4) voxels one pass covers (keeps the serial
// loop short at the coarse, wide-channel levels)
static inline int in_chunk(int C) { return C >= 4 ? (BLK / (C >> 2)) * 32 : BLK * 32; }   // C % 4 != 0 is refused by the callers

// d_y of a level's output block formed on the fly (MODE 2 / POOL = true below): the block's output y went to the level's
// consumers (moving half: add_a, fixed half: add_b, either may be null) and, through AvgPool3d(2), to the next level (gy), so
//   d_y[b][v] = gy[b][v / 2] * 0.125 + add[b][v]
// -- the arithmetic of avgpool2_bwd_kernel, which used to write this tensor for the two passes below to read back.
struct PoolSrc { const float* gy; const float* add_a; const float* add_b; int Bh, D, H, W; };
__device__ __forceinline__ void pool_dy_addr(const PoolSrc& p, int b, int64_t v, int64_t V, int C, int g, const float*& pg,
                                             const float*& pa) {
  const unsigned uv = (unsigned)v, q1 = uv / (unsigned)p.W, q2 = q1 / (unsigned)p.H;          // (v < 2^31: checked by the host)
  const int xi = (int)(uv - q1 * (unsigned)p.W), yi = (int)(q1 - q2 * (unsigned)p.H), zi = (int)q2;
  const int h = p.H >> 1, w = p.W >> 1, d = p.D >> 1;
  pg = p.gy + ((((int64_t)b * d + (zi >> 1)) * h + (yi >> 1)) * w + (xi >> 1)) * C + g * 4;
  const float* base = b < p.Bh ? p.add_a : p.add_b;
  pa = base ? base + ((int64_t)(b < p.Bh ? b : b - p.Bh) * V + v) * C + g * 4 : nullptr;
}
__device__ __forceinline__ float4 pool_dy_value(const float4 gv, const float4 av, bool has_add) {
  float4 r = gv;
  r.x *= 0.125f; r.y *= 0.125f; r.z *= 0.125f; r.w *= 0.125f;
  if (has_add) { r.x += av.x; r.y += av.y; r.z += av.z; r.w += av.w; }
  return r;
}

// MODE 0: (sum x, sum x^2)        MODE 1: (sum g, sum g*xhat), g = dy * lrelu'(xhat)        MODE 2: MODE 1 with d_y from PoolSrc
template <int MODE>
__global__ __launch_bounds__(BLK) void in_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ part, int64_t V, int C, int chunk,
                                                         const PoolSrc ps = PoolSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0}) {
  __shared__ float red[BLK * 8];
  const int G = C >> 2;                        // float4 groups per voxel
  const int VPB = BLK / G;                     // voxels per pass
  const int b = blockIdx.y;
  const int g = threadIdx.x % G, vl = threadIdx.x / G;
  const bool active = vl < VPB;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
  if (MODE >= 1 && active) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { mu[c] = mean[b * C + g * 4 + c]; rs[c] = rstd[b * C + g * 4 + c]; }
  }
  const int64_t v0 = (int64_t)blockIdx.x * chunk;
  const int64_t v1 = v0 + chunk < V ? v0 + chunk : V;
  if (active) {
    auto acc1 = [&](const float4 xv, const float4 gv) {
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[c] += xs[c]; q[c] = fmaf(xs[c], xs[c], q[c]); }
      } else {
        const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float xh = (xs[c] - mu[c]) * rs[c];
          const float gg = gs[c] * (xh > 0.f ? 1.f : LRELU_SLOPE);
          a[c] += gg; q[c] = fmaf(gg, xh, q[c]);
        }
      }
    };
    // the loads of PU passes are issued together, the sums keep their order (bit-identical to the one-pass-per-trip loop)
    constexpr int PU = MODE == 0 ? 4 : 2;
    int64_t v = v0 + vl;
    for (; v + (PU - 1) * VPB < v1; v += PU * VPB) {
      float4 xv[PU], gv[PU], av[PU];
      bool ha[PU];
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int64_t off = ((int64_t)b * V + v + u * VPB) * C + g * 4;
        xv[u] = *reinterpret_cast<const float4*>(x + off);
        gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        av[u] = gv[u];
        ha[u] = false;
        if (MODE == 1) gv[u] = *reinterpret_cast<const float4*>(dy + off);
        if (MODE == 2) {
          const float *pg, *pa;
          pool_dy_addr(ps, b, v + u * VPB, V, C, g, pg, pa);
          gv[u] = *reinterpret_cast<const float4*>(pg);
          ha[u] = pa != nullptr;                                     // (uniform per sample half)
          av[u] = *reinterpret_cast<const float4*>(ha[u] ? pa : pg);
        }
      }
      if (MODE == 2) {
#pragma unroll
        for (int u = 0; u < PU; ++u) asm volatile("" : "+v"(xv[u].x), "+v"(gv[u].x), "+v"(av[u].x));
      }
#pragma unroll
      for (int u = 0; u < PU; ++u) acc1(xv[u], MODE == 2 ? pool_dy_value(gv[u], av[u], ha[u]) : gv[u]);
    }
    for (; v < v1; v += VPB) {
      const int64_t off = ((int64_t)b * V + v) * C + g * 4;
      const float4 xv = *reinterpret_cast<const float4*>(x + off);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) gv = *reinterpret_cast<const float4*>(dy + off);
      if (MODE == 2) {
        const float *pg, *pa;
        pool_dy_addr(ps, b, v, V, C, g, pg, pa);
        const float4 g4 = *reinterpret_cast<const float4*>(pg);
        gv = pool_dy_value(g4, pa ? *reinterpret_cast<const float4*>(pa) : g4, pa != nullptr);
      }
      acc1(xv, gv);
    }
  }
  if ((G & (G - 1)) == 0 && G <= 64) {
    // channel group = lane % G: a fixed xor-shuffle tree over the lanes of equal group, then the 4 waves through LDS
    // (the former tail -- G threads walking BLK/G entries each, serially -- cost 40 us of a 157 us kernel at C = 8)
    for (int o = G; o < 64; o <<= 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] += __shfl_xor(a[c], o, 64); q[c] += __shfl_xor(q[c], o, 64); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < G) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { red[(wave * G + lane) * 8 + c] = a[c]; red[(wave * G + lane) * 8 + 4 + c] = q[c]; }
    }
    __syncthreads();
    if (threadIdx.x < G) {
      double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int w = 0; w < BLK / 64; ++w) {
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c] += (double)red[(w * G + threadIdx.x) * 8 + c];
      }
      float* p = part + (((int64_t)b * gridDim.x + blockIdx.x) * C + threadIdx.x * 4) * 2;
#pragma unroll
      for (int c = 0; c < 4; ++c) { p[c * 2] = (float)s[c]; p[c * 2 + 1] = (float)s[4 + c]; }
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) { red[threadIdx.x * 8 + c] = a[c]; red[threadIdx.x * 8 + 4 + c] = q[c]; }
  __syncthreads();
  if (threadIdx.x < G) {
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // fp64: no coherent rounding when the lanes' sums are alike
    for (int j = 0; j < VPB; ++j) {
#pragma unroll
      for (int c = 0; c < 8; ++c) s[c] += (double)red[(j * G + threadIdx.x) * 8 + c];
    }
    float* p = part + (((int64_t)b * gridDim.x + blockIdx.x) * C + threadIdx.x * 4) * 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) { p[c * 2] = (float)s[c]; p[c * 2 + 1] = (float)s[4 + c]; }
  }
}

// MODE 0: -> mean, rstd.   MODE 1: -> (sum g)/V, (sum g*xhat)/V in out0/out1.
template <int MODE>
__global__ __launch_bounds__(64) void in_finalize_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                         float* __restrict__ out1, int64_t V, int C, int nchunk, float eps,
                                                         float* __restrict__ amax_zero = nullptr) {
  const int b = blockIdx.y, c = blockIdx.x;       // one wave per (b,c): fixed assignment + fixed tree, fp64
  if (amax_zero && b == 0 && c == 0) amax_zero[threadIdx.x * MODET_AMAX_STRIDE] = 0.f;   // (64 threads = MODET_AMAX_SLOTS) the apply pass maxes into them
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nchunk; i += 64) {
    const float* p = part + (((int64_t)b * nchunk + i) * C + c) * 2;
    s += (double)p[0]; q += (double)p[1];
  }
  s = wave_sum_d(s); q = wave_sum_d(q);
  if (threadIdx.x != 0) return;
  if (MODE == 0) {
    const double m = s / (double)V;
    double var = q / (double)V - m * m;
    if (var < 0.0) var = 0.0;
    out0[b * C + c] = (float)m;
    out1[b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    out0[b * C + c] = (float)(s / (double)V);
    out1[b * C + c] = (float)(q / (double)V);
  }
}

// statistics from the conv epilogue: stats = [B][C] shift K, then partial rows [(b*rows_per_b + r)][C][2] of
// sum(y - K), sum((y - K)^2) (one row per workgroup of the conv; see ConvIn in conv3d.hip for why they are shifted):
// one workgroup per sample, coalesced fixed-order fp64 column sums, then mean = K + s1/V, var = s2/V - (s1/V)^2 in fp64
constexpr int IN_FIN_COLS = 8;                        // columns (= 8 channels x 2 sums) per finalize workgroup
__global__ __launch_bounds__(256) void in_rows_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ rows,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int64_t V,
                                                               int C, int64_t rows_per_b, float eps) {
  __shared__ double sm[256];
  __shared__ double tot[256];
  // grid (B, column groups): a workgroup sums IN_FIN_COLS of the 2 C columns (one workgroup per sample walked thousands of
  // rows with 4 row lanes at 64 columns: 20-40 us of dependent L2 latency per launch, 14 launches per train step)
  const int b = blockIdx.x, c0 = blockIdx.y * IN_FIN_COLS;
  const int ncol = 2 * C - c0 < IN_FIN_COLS ? 2 * C - c0 : IN_FIN_COLS;
  block_colsum_256(rows + (int64_t)b * rows_per_b * 2 * C + c0, 0, rows_per_b, ncol, tot, sm, 2 * C);
  __syncthreads();
  if ((int)threadIdx.x < ncol / 2) {
    const int cl = threadIdx.x, c = c0 / 2 + cl;
    const double m = tot[2 * cl] / (double)V;
    double var = tot[2 * cl + 1] / (double)V - m * m;
    if (var < 0.0) var = 0.0;
    mean[b * C + c] = (float)((double)stats[b * C + c] + m);
    rstd[b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// the same column sums for the InstanceNorm BACKWARD rows a data-gradient kernel left (sum g, sum g*xhat): -> means over V
__global__ __launch_bounds__(256) void in_rows_finalize_bwd_kernel(const float* __restrict__ rows, float* __restrict__ s1,
                                                                   float* __restrict__ s2, int64_t V, int C, int64_t rows_per_b,
                                                                   float* __restrict__ amax_zero = nullptr) {
  __shared__ double sm[256];
  __shared__ double tot[256];
  const int b = blockIdx.x, c0 = blockIdx.y * IN_FIN_COLS;
  const int ncol = 2 * C - c0 < IN_FIN_COLS ? 2 * C - c0 : IN_FIN_COLS;
  if (amax_zero && b == 0 && blockIdx.y == 0 && threadIdx.x < MODET_AMAX_SLOTS) amax_zero[threadIdx.x * MODET_AMAX_STRIDE] = 0.f;
  block_colsum_256(rows + (int64_t)b * rows_per_b * 2 * C + c0, 0, rows_per_b, ncol, tot, sm, 2 * C);
  __syncthreads();
  if ((int)threadIdx.x < ncol / 2) {
    const int cl = threadIdx.x, c = c0 / 2 + cl;
    s1[b * C + c] = (float)(tot[2 * cl] / (double)V);
    s2[b * C + c] = (float)(tot[2 * cl + 1] / (double)V);
  }
}

// stage 1 for statistics with one row per OUTPUT TILE (bf16 convs: thousands of rows per sample): IN_SLICES workgroups per
// sample sum a slice of the rows each (fixed order, fp64) into one row of the buffer's tail; the finalize then reads
// IN_SLICES rows.  (One workgroup walking 9 600 rows took 71 us per level-1 layer.)
constexpr int IN_SLICES = 64;
__global__ __launch_bounds__(256) void in_rows_slice_kernel(const float* __restrict__ rows, float* __restrict__ tail, int C,
                                                            int64_t rows_per_b) {
  __shared__ double sm[256];
  __shared__ double tot[256];
  const int b = blockIdx.y, sl = blockIdx.x;
  const int64_t per = (rows_per_b + IN_SLICES - 1) / IN_SLICES;
  const int64_t r0 = sl * per < rows_per_b ? sl * per : rows_per_b;                 // empty slices sum nothing
  const int64_t r1 = r0 + per < rows_per_b ? r0 + per : rows_per_b;
  block_colsum_256(rows + (int64_t)b * rows_per_b * 2 * C, r0, r1, 2 * C, tot, sm);
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) tail[((int64_t)b * IN_SLICES + sl) * 2 * C + threadIdx.x] = (float)tot[threadIdx.x];
}

// (channel group, sample) of float4 element i.  32-bit arithmetic whenever the tensor allows it: a 64-bit division by a
// run-time value is a ~100-instruction software routine, and these kernels have ~10 instructions of real work per element.
__device__ __forceinline__ void in_elem(int64_t i, int G, int64_t V, bool small, int& g, int& b) {
  if (small) {
    const unsigned u = (unsigned)i, q = u / (unsigned)G;
    g = (int)(u - q * (unsigned)G);
    b = (int)(q / (unsigned)V);
  } else {
    g = (int)(i % G);
    b = (int)((i / G) / V);
  }
}
constexpr int IN_ILP = 4;      // float4 elements per thread and trip, all loads issued before the first use

__global__ __launch_bounds__(BLK) void in_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       int64_t V, int C, int64_t total4) {
  const int G = C >> 2;
  const bool small = total4 < (1ll << 31) && V < (1ll << 31);
  const int64_t stride = (int64_t)gridDim.x * BLK;
  auto one = [&](int64_t i, const float4 xv, const float4 m, const float4 r) {
    float4 o;
    o.x = lrelu((xv.x - m.x) * r.x); o.y = lrelu((xv.y - m.y) * r.y);
    o.z = lrelu((xv.z - m.z) * r.z); o.w = lrelu((xv.w - m.w) * r.w);
    reinterpret_cast<float4*>(y)[i] = o;
  };
  auto cidx = [&](int64_t i) { int g, b; in_elem(i, G, V, small, g, b); return b * C + g * 4; };
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  for (; i + (IN_ILP - 1) * stride < total4; i += IN_ILP * stride) {
    float4 xv[IN_ILP], m[IN_ILP], r[IN_ILP];
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) xv[u] = reinterpret_cast<const float4*>(x)[i + u * stride];
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) {
      const int ci = cidx(i + u * stride);
      m[u] = *reinterpret_cast<const float4*>(mean + ci);
      r[u] = *reinterpret_cast<const float4*>(rstd + ci);
    }
    // every load of the trip in flight before the first use (hipcc otherwise waits per element for its mean / rstd)
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) asm volatile("" : "+v"(m[u].x), "+v"(r[u].x), "+v"(xv[u].x));
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) one(i + u * stride, xv[u], m[u], r[u]);
  }
  for (; i < total4; i += stride) {
    const int ci = cidx(i);
    one(i, reinterpret_cast<const float4*>(x)[i], *reinterpret_cast<const float4*>(mean + ci),
        *reinterpret_cast<const float4*>(rstd + ci));
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat));  POOL: d_y is formed on the fly from PoolSrc (see above) instead of read.
// amax != null: max |dx| over the whole tensor is left there (MODET_AMAX_SLOTS slots, one integer atomic max per wave on the
// float's bits; the slots were zeroed by the finalize kernel in front) -- the scale with which the convolutions that consume dx split it into f16
// pieces (modet_conv3d_bwd_data_amax / _bwd_weight_amax).
template <bool POOL>
__global__ __launch_bounds__(BLK) void in_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           float* __restrict__ dx, int64_t V, int C, int64_t total4,
                                                           const PoolSrc ps = PoolSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0},
                                                           float* __restrict__ amax = nullptr) {
  const int G = C >> 2;
  float tmax = 0.f;
  const bool small = total4 < (1ll << 31) && V < (1ll << 31);
  const int64_t stride = (int64_t)gridDim.x * BLK;
  auto one = [&](int64_t i, const float4 xv, const float4 gv, const float4 m4, const float4 r4, const float4 a4,
                 const float4 b4) {
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
    const float ms[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
    const float s1v[4] = {a4.x, a4.y, a4.z, a4.w}, s2v[4] = {b4.x, b4.y, b4.z, b4.w};
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float r = rs[c];
      const float xh = (xs[c] - ms[c]) * r;
      const float gg = gs[c] * (xh > 0.f ? 1.f : LRELU_SLOPE);
      o[c] = r * (gg - s1v[c] - xh * s2v[c]);
    }
    tmax = fmaxf(fmaxf(tmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
  };
  auto cidx = [&](int64_t i) { int g, b; in_elem(i, G, V, small, g, b); return b * C + g * 4; };
  constexpr int ILP = 2;       // two tensors are read: 4 float4 loads in flight per thread
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  for (; i + (ILP - 1) * stride < total4; i += ILP * stride) {
    float4 xv[ILP], gv[ILP], m4[ILP], r4[ILP], a4[ILP], b4[ILP], pv[ILP];
    bool ha[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      xv[u] = reinterpret_cast<const float4*>(x)[i + u * stride];
      if (!POOL) {
        gv[u] = reinterpret_cast<const float4*>(dy)[i + u * stride];
      } else {
        int g, b;
        in_elem(i + u * stride, G, V, small, g, b);
        const int64_t v = (i + u * stride) / G - (int64_t)b * V;
        const float *pg, *pa;
        pool_dy_addr(ps, b, v, V, C, g, pg, pa);
        gv[u] = *reinterpret_cast<const float4*>(pg);
        ha[u] = pa != nullptr;
        pv[u] = *reinterpret_cast<const float4*>(ha[u] ? pa : pg);
      }
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int ci = cidx(i + u * stride);                 // 4 channels of one sample: contiguous in all four arrays
      m4[u] = *reinterpret_cast<const float4*>(mean + ci); r4[u] = *reinterpret_cast<const float4*>(rstd + ci);
      a4[u] = *reinterpret_cast<const float4*>(s1 + ci);   b4[u] = *reinterpret_cast<const float4*>(s2 + ci);
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      asm volatile("" : "+v"(xv[u].x), "+v"(gv[u].x), "+v"(m4[u].x), "+v"(r4[u].x), "+v"(a4[u].x), "+v"(b4[u].x));
      if (POOL) asm volatile("" : "+v"(pv[u].x));
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u)
      one(i + u * stride, xv[u], POOL ? pool_dy_value(gv[u], pv[u], ha[u]) : gv[u], m4[u], r4[u], a4[u], b4[u]);
  }
  for (; i < total4; i += stride) {
    const int ci = cidx(i);
    float4 gv;
    if (!POOL) {
      gv = reinterpret_cast<const float4*>(dy)[i];
    } else {
      int g, b;
      in_elem(i, G, V, small, g, b);
      const float *pg, *pa;
      pool_dy_addr(ps, b, i / G - (int64_t)b * V, V, C, g, pg, pa);
      const float4 g4 = *reinterpret_cast<const float4*>(pg);
      gv = pool_dy_value(g4, pa ? *reinterpret_cast<const float4*>(pa) : g4, pa != nullptr);
    }
    one(i, reinterpret_cast<const float4*>(x)[i], gv,
        *reinterpret_cast<const float4*>(mean + ci), *reinterpret_cast<const float4*>(rstd + ci),
        *reinterpret_cast<const float4*>(s1 + ci), *reinterpret_cast<const float4*>(s2 + ci));
  }
  if (amax) {                                            // (uniform branch; non-negative floats order like their bit patterns)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
    // all waves finish together: maxing into ONE address serialises them in a single L2 channel (measured +1.2 ms per train
    // step; +0.5 with a read in front of the atomic) -- a workgroup's waves use slot blockIdx % 64, the slots 128 bytes apart,
    // and only send the atomic when it would raise the slot (device-scope load, past the non-coherent L1)
    if ((threadIdx.x & 63) == 0) {
      unsigned* slot = reinterpret_cast<unsigned*>(amax) + (blockIdx.x % MODET_AMAX_SLOTS) * MODET_AMAX_STRIDE;
      const unsigned mine = __float_as_uint(tmax);
      if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
    }
  }
}

__global__ __launch_bounds__(BLK) void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK)
    dx[i] = dy[i] * (y[i] > 0.f ? 1.f : LRELU_SLOPE);
}

__global__ __launch_bounds__(BLK) void scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                    float* __restrict__ y, int64_t n) {
  const float f = s[0];
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) y[i] = x[i] * f;
}

// ---- AvgPool3d(2): (B,D,H,W,C) -> (B,D/2,H/2,W/2,C), float4 channel groups
template <bool X16>                                  // X16: x holds bf16 (widened on load), y is fp32 either way
__global__ __launch_bounds__(BLK) void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int D,
                                                           int H, int W, int C, int64_t total4) {
  const int G = C >> 2, d = D / 2, h = H / 2, w = W / 2;
  const bool small = total4 < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total4; i += (int64_t)gridDim.x * BLK) {
    int g, xo, yo, zo;
    int64_t b;
    if (small) {                 // 32-bit index arithmetic (64-bit div/mod: ~100 instructions each, five per element)
      unsigned t = (unsigned)i, q = t / (unsigned)G;
      g = (int)(t - q * (unsigned)G); t = q;
      q = t / (unsigned)w; xo = (int)(t - q * (unsigned)w); t = q;
      q = t / (unsigned)h; yo = (int)(t - q * (unsigned)h); t = q;
      q = t / (unsigned)d; zo = (int)(t - q * (unsigned)d);
      b = q;
    } else {
      g = (int)(i % G);
      int64_t t = i / G;
      xo = (int)(t % w); t /= w;
      yo = (int)(t % h); t /= h;
      zo = (int)(t % d);
      b = t / d;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int64_t off = (((b * D + 2 * zo + dz) * H + 2 * yo + dy) * W + 2 * xo + dx) * C + g * 4;
          float4 v;
          if constexpr (X16) {
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + off);
            v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xffff0000u));
          } else {
            v = *reinterpret_cast<const float4*>(x + off);
          }
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    s.x *= 0.125f; s.y *= 0.125f; s.z *= 0.125f; s.w *= 0.125f;
    reinterpret_cast<float4*>(y)[i] = s;
  }
}

// InstanceNorm apply + LeakyReLU + AvgPool3d(2) in one pass (the last block of an encoder level: its output goes to the
// level's consumers AND, pooled, to the next level): thread = (pooled voxel, 4 channels) reads its 2x2x2 raw voxels (all
// eight loads in flight), writes the eight normalised voxels and their mean.  y and the pooled sum use the arithmetic and the
// (dz, dy, dx) order of in_apply_kernel / avgpool2_fwd_kernel: bit-identical to the two-pass form, which re-read y (8 B/element).
__global__ __launch_bounds__(BLK) void in_apply_pool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ pooled, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int D, int H, int W, int C,
                                                            int64_t total4) {
  const int G = C >> 2, d = D / 2, h = H / 2, w = W / 2;
  const bool small = total4 < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total4; i += (int64_t)gridDim.x * BLK) {
    int g, xo, yo, zo;
    int64_t b;
    if (small) {
      unsigned t = (unsigned)i, q = t / (unsigned)G;
      g = (int)(t - q * (unsigned)G); t = q;
      q = t / (unsigned)w; xo = (int)(t - q * (unsigned)w); t = q;
      q = t / (unsigned)h; yo = (int)(t - q * (unsigned)h); t = q;
      q = t / (unsigned)d; zo = (int)(t - q * (unsigned)d);
      b = q;
    } else {
      g = (int)(i % G);
      int64_t t = i / G;
      xo = (int)(t % w); t /= w;
      yo = (int)(t % h); t /= h;
      zo = (int)(t % d);
      b = t / d;
    }
    const float4 m = *reinterpret_cast<const float4*>(mean + b * C + g * 4);
    const float4 r = *reinterpret_cast<const float4*>(rstd + b * C + g * 4);
    float4 v[8];
    int64_t off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      off[k] = (((b * D + 2 * zo + (k >> 2)) * H + 2 * yo + ((k >> 1) & 1)) * W + 2 * xo + (k & 1)) * C + g * 4;
      v[k] = *reinterpret_cast<const float4*>(x + off[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float4 o;
      o.x = lrelu((v[k].x - m.x) * r.x); o.y = lrelu((v[k].y - m.y) * r.y);
      o.z = lrelu((v[k].z - m.z) * r.z); o.w = lrelu((v[k].w - m.w) * r.w);
      *reinterpret_cast<float4*>(y + off[k]) = o;
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    s.x *= 0.125f; s.y *= 0.125f; s.z *= 0.125f; s.w *= 0.125f;
    reinterpret_cast<float4*>(pooled)[i] = s;
  }
}

// dx = upsample(dy)/8 (+ addend): the optional addend is the gradient arriving on the un-pooled branch of the same
// tensor, so "pool backward" and autograd's "sum of the two consumers" become one pass
__global__ __launch_bounds__(BLK) void avgpool2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ addend,
                                                           float* __restrict__ dx, int D, int H, int W, int C,
                                                           int64_t total4) {
  const int G = C >> 2, h = H / 2, w = W / 2, d = D / 2;
  const bool small = total4 < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total4; i += (int64_t)gridDim.x * BLK) {
    int g, xi, yi, zi;
    int64_t b;
    if (small) {
      unsigned t = (unsigned)i, q = t / (unsigned)G;
      g = (int)(t - q * (unsigned)G); t = q;
      q = t / (unsigned)W; xi = (int)(t - q * (unsigned)W); t = q;
      q = t / (unsigned)H; yi = (int)(t - q * (unsigned)H); t = q;
      q = t / (unsigned)D; zi = (int)(t - q * (unsigned)D);
      b = q;
    } else {
      g = (int)(i % G);
      int64_t t = i / G;
      xi = (int)(t % W); t /= W;
      yi = (int)(t % H); t /= H;
      zi = (int)(t % D);
      b = t / D;
    }
    const int64_t off = (((b * d + zi / 2) * h + yi / 2) * w + xi / 2) * C + g * 4;
    float4 v = *reinterpret_cast<const float4*>(dy + off);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (addend) a = reinterpret_cast<const float4*>(addend)[i];              // (uniform) both loads in flight together
    asm volatile("" : "+v"(v.x), "+v"(a.x));
    v.x *= 0.125f; v.y *= 0.125f; v.z *= 0.125f; v.w *= 0.125f;
    if (addend) { v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    reinterpret_cast<float4*>(dx)[i] = v;
  }
}

// ---- Adam(amsgrad=True): single-tensor update of torch.optim.Adam (bias corrections from `step`)
__global__ __launch_bounds__(BLK) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, int64_t n, float step_size, float beta1,
                                                   float beta2, float eps, float inv_sqrt_bc2, float grad_scale) {
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
    const float gi = g[i] * grad_scale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;      // exp_avg.lerp_(grad, 1-beta1)
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    const float vm = fmaxf(vmax[i], vi);
    m[i] = mi; v[i] = vi; vmax[i] = vm;
    const float denom = sqrtf(vm) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}

// ------------------------------------------------------------------------------------------------ bf16 storage (cfg 5)
// InstanceNorm + LeakyReLU on bf16-stored conv outputs (csrc/conv3d_bf16.hip): 8 channels = 16 bytes per thread,
// all arithmetic in fp32, statistics (mean / rstd, the backward sums) fp32 as in the fp32 path.
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned bf_pack(float a, float b) {            // round to nearest even (v_cvt_pk_bf16_f32)
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ void bf_unpack8(const uint4 v, float (&f)[8]) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
template <bool BF>
__device__ __forceinline__ void load8(const void* base, int64_t i8, float (&f)[8]) {      // i8: index in units of 8 elements
  if (BF) {
    bf_unpack8(reinterpret_cast<const uint4*>(base)[i8], f);
  } else {
    const float4 a = reinterpret_cast<const float4*>(base)[2 * i8], b = reinterpret_cast<const float4*>(base)[2 * i8 + 1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
}
template <bool BF>
__device__ __forceinline__ void store8(void* base, int64_t i8, const float (&f)[8]) {
  if (BF) {
    reinterpret_cast<uint4*>(base)[i8] = make_uint4(bf_pack(f[0], f[1]), bf_pack(f[2], f[3]), bf_pack(f[4], f[5]), bf_pack(f[6], f[7]));
  } else {
    reinterpret_cast<float4*>(base)[2 * i8] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4*>(base)[2 * i8 + 1] = make_float4(f[4], f[5], f[6], f[7]);
  }
}

// raw form of load8: the load(s) only, so that several elements' loads can be issued (and fenced) before the first unpack
struct Raw8 { uint4 a, b; };
template <bool BF>
__device__ __forceinline__ Raw8 load8_raw(const void* base, int64_t i8) {
  Raw8 r;
  if (BF) { r.a = reinterpret_cast<const uint4*>(base)[i8]; r.b = make_uint4(0u, 0u, 0u, 0u); }
  else { r.a = reinterpret_cast<const uint4*>(base)[2 * i8]; r.b = reinterpret_cast<const uint4*>(base)[2 * i8 + 1]; }
  return r;
}
template <bool BF>
__device__ __forceinline__ void raw8_fence(Raw8& r) {
  if (BF) asm volatile("" : "+v"(r.a.x)); else asm volatile("" : "+v"(r.a.x), "+v"(r.b.x));
}
template <bool BF>
__device__ __forceinline__ void unpack8(const Raw8& r, float (&f)[8]) {
  if (BF) {
    bf_unpack8(r.a, f);
  } else {
    f[0] = __uint_as_float(r.a.x); f[1] = __uint_as_float(r.a.y); f[2] = __uint_as_float(r.a.z); f[3] = __uint_as_float(r.a.w);
    f[4] = __uint_as_float(r.b.x); f[5] = __uint_as_float(r.b.y); f[6] = __uint_as_float(r.b.z); f[7] = __uint_as_float(r.b.w);
  }
}
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// voxels per partial-sum workgroup: 16 passes (twice the workgroups of the fp32 kernel per byte: a pass moves half the bytes)
static inline int in_chunk8(int C) { return C >= 8 ? (BLK / (C >> 3)) * 16 : BLK * 16; }   // C % 8 != 0 is refused by the callers

// d_y of a level's output block formed on the fly from PoolSrc (fp32 gradients of the level's consumers + the pooled gradient),
// 8 channels per thread: DYM = 2 of the two backward kernels below (0: d_y fp32, 1: d_y bf16)
struct PoolDy8 { Raw8 g, a; bool has; };
__device__ __forceinline__ PoolDy8 pool_dy8_load(const PoolSrc& ps, int b, int64_t v, int64_t V, int C, int g8) {
  const float *pg, *pa;
  pool_dy_addr(ps, b, v, V, C, g8 * 2, pg, pa);
  PoolDy8 r;
  r.has = pa != nullptr;
  r.g = load8_raw<false>(pg, 0);
  r.a = load8_raw<false>(r.has ? pa : pg, 0);
  return r;
}
__device__ __forceinline__ void pool_dy8_fence(PoolDy8& r) { raw8_fence<false>(r.g); raw8_fence<false>(r.a); }
__device__ __forceinline__ void pool_dy8_unpack(const PoolDy8& r, float (&f)[8]) {
  float gv[8], av[8];
  unpack8<false>(r.g, gv);
  unpack8<false>(r.a, av);
#pragma unroll
  for (int c = 0; c < 8; ++c) f[c] = r.has ? gv[c] * 0.125f + av[c] : gv[c] * 0.125f;      // (as pool_dy_value: mul, then add)
}

template <bool OUT_BF>
__global__ __launch_bounds__(BLK) void in_apply_bf16_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int64_t V, int C, int64_t total8) {
  const int G = C >> 3;
  const bool small = total8 < (1ll << 31) && V < (1ll << 31);
  const int64_t stride = (int64_t)gridDim.x * BLK;
  auto one = [&](int64_t i, const Raw8& rx) {
    int g, b;
    in_elem(i, G, V, small, g, b);
    float xv[8], o[8], m[8], r[8];
    unpack8<true>(rx, xv);
    ld8f(mean + b * C + g * 8, m);
    ld8f(rstd + b * C + g * 8, r);
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = lrelu((xv[c] - m[c]) * r[c]);
    store8<OUT_BF>(y, i, o);
  };
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  for (; i + (IN_ILP - 1) * stride < total8; i += IN_ILP * stride) {      // IN_ILP loads in flight per thread
    Raw8 rx[IN_ILP];
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) rx[u] = load8_raw<true>(x, i + u * stride);
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) raw8_fence<true>(rx[u]);
#pragma unroll
    for (int u = 0; u < IN_ILP; ++u) one(i + u * stride, rx[u]);
  }
  for (; i < total8; i += stride) one(i, load8_raw<true>(x, i));
}

// in_apply_bf16_kernel<true> + AvgPool3d(2) in one pass (BASELINE.json configs[4], a level's output block whose features are stored
// as bf16): thread = (pooled voxel, 8 channels) reads its 2x2x2 raw voxels (all eight 16-byte loads in flight), writes the eight
// normalised voxels as bf16 and their mean -- formed from the fp32 values IN FRONT of the rounding -- as fp32: pooling the
// rounded features instead was measured to cost the gradient (relative L2 0.12 -> 0.19 at 64^3, tools/micro/emul_t1.py)
__global__ __launch_bounds__(BLK) void in_apply_pool_bf16_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                                 float* __restrict__ pooled, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int D, int H, int W, int C,
                                                                 int64_t total8) {
  const int G = C >> 3, d = D / 2, h = H / 2, w = W / 2;
  const bool small = total8 < (1ll << 31);
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total8; i += (int64_t)gridDim.x * BLK) {
    int g, xo, yo, zo;
    int64_t b;
    if (small) {
      unsigned t = (unsigned)i, q = t / (unsigned)G;
      g = (int)(t - q * (unsigned)G); t = q;
      q = t / (unsigned)w; xo = (int)(t - q * (unsigned)w); t = q;
      q = t / (unsigned)h; yo = (int)(t - q * (unsigned)h); t = q;
      q = t / (unsigned)d; zo = (int)(t - q * (unsigned)d);
      b = q;
    } else {
      g = (int)(i % G);
      int64_t t = i / G;
      xo = (int)(t % w); t /= w;
      yo = (int)(t % h); t /= h;
      zo = (int)(t % d);
      b = t / d;
    }
    float m[8], r[8];
    ld8f(mean + b * C + g * 8, m);
    ld8f(rstd + b * C + g * 8, r);
    Raw8 rv[8];
    int64_t i8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      i8[k] = (((b * D + 2 * zo + (k >> 2)) * H + 2 * yo + ((k >> 1) & 1)) * W + 2 * xo + (k & 1)) * G + g;
      rv[k] = load8_raw<true>(x, i8[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) raw8_fence<true>(rv[k]);
    float s[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float xv[8], o[8];
      unpack8<true>(rv[k], xv);
#pragma unroll
      for (int c = 0; c < 8; ++c) { o[c] = lrelu((xv[c] - m[c]) * r[c]); s[c] += o[c]; }
      store8<true>(y, i8[k], o);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] *= 0.125f;
    store8<false>(pooled, i, s);
  }
}

// (sum g, sum g*xhat), g = dy * lrelu'(xhat); x is the bf16 raw conv output
template <int DYM>
__global__ __launch_bounds__(BLK) void in_partial_bf16_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              float* __restrict__ part, int64_t V, int C, int chunk,
                                                              const PoolSrc ps = PoolSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0}) {
  constexpr bool DY_BF = DYM == 1, POOL = DYM == 2;
  __shared__ float red[BLK * 16];
  const int G = C >> 3, VPB = BLK / G;
  const int b = blockIdx.y;
  const int g = threadIdx.x % G, vl = threadIdx.x / G;
  const bool active = vl < VPB;
  float a[8], q[8], mu[8], rs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { a[c] = 0.f; q[c] = 0.f; mu[c] = 0.f; rs[c] = 1.f; }
  if (active) {
#pragma unroll
    for (int c = 0; c < 8; ++c) { mu[c] = mean[b * C + g * 8 + c]; rs[c] = rstd[b * C + g * 8 + c]; }
  }
  const int64_t v0 = (int64_t)blockIdx.x * chunk;
  const int64_t v1 = v0 + chunk < V ? v0 + chunk : V;
  if (active) {
    auto accg = [&](const Raw8& rx, const float (&gs)[8]) {
      float xs[8];
      unpack8<true>(rx, xs);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float xh = (xs[c] - mu[c]) * rs[c];
        const float gg = gs[c] * (xh > 0.f ? 1.f : LRELU_SLOPE);
        a[c] += gg; q[c] = fmaf(gg, xh, q[c]);
      }
    };
    auto acc1 = [&](const Raw8& rx, const Raw8& rg) {
      float gs[8];
      unpack8<DY_BF>(rg, gs);
      accg(rx, gs);
    };
    // the loads of PU passes are issued and fenced together, the sums keep their order (bit-identical)
    constexpr int PU = POOL ? 2 : 4;
    int64_t v = v0 + vl;
    for (; v + (PU - 1) * VPB < v1; v += PU * VPB) {
      Raw8 rx[PU], rg[PU];
      PoolDy8 pd[PU];
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int64_t i8 = ((int64_t)b * V + v + u * VPB) * G + g;
        rx[u] = load8_raw<true>(x, i8);
        if constexpr (POOL) pd[u] = pool_dy8_load(ps, b, v + u * VPB, V, C, g);
        else rg[u] = load8_raw<DY_BF>(dy, i8);
      }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        raw8_fence<true>(rx[u]);
        if constexpr (POOL) pool_dy8_fence(pd[u]); else raw8_fence<DY_BF>(rg[u]);
      }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        if constexpr (POOL) { float gs[8]; pool_dy8_unpack(pd[u], gs); accg(rx[u], gs); }
        else acc1(rx[u], rg[u]);
      }
    }
    for (; v < v1; v += VPB) {
      const int64_t i8 = ((int64_t)b * V + v) * G + g;
      if constexpr (POOL) { float gs[8]; pool_dy8_unpack(pool_dy8_load(ps, b, v, V, C, g), gs); accg(load8_raw<true>(x, i8), gs); }
      else acc1(load8_raw<true>(x, i8), load8_raw<DY_BF>(dy, i8));
    }
  }
  if ((G & (G - 1)) == 0 && G <= 64) {                   // see in_partial_kernel: shuffle tree, then the 4 waves through LDS
    for (int o = G; o < 64; o <<= 1) {
#pragma unroll
      for (int c = 0; c < 8; ++c) { a[c] += __shfl_xor(a[c], o, 64); q[c] += __shfl_xor(q[c], o, 64); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < G) {
#pragma unroll
      for (int c = 0; c < 8; ++c) { red[(wave * G + lane) * 16 + c] = a[c]; red[(wave * G + lane) * 16 + 8 + c] = q[c]; }
    }
    __syncthreads();
    if (threadIdx.x < G) {
      double s[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) s[c] = 0.0;
      for (int w = 0; w < BLK / 64; ++w) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c] += (double)red[(w * G + threadIdx.x) * 16 + c];
      }
      float* p = part + (((int64_t)b * gridDim.x + blockIdx.x) * C + threadIdx.x * 8) * 2;
#pragma unroll
      for (int c = 0; c < 8; ++c) { p[c * 2] = (float)s[c]; p[c * 2 + 1] = (float)s[8 + c]; }
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) { red[threadIdx.x * 16 + c] = a[c]; red[threadIdx.x * 16 + 8 + c] = q[c]; }
  __syncthreads();
  if (threadIdx.x < G) {
    double s[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s[c] = 0.0;
    for (int j = 0; j < VPB; ++j) {
#pragma unroll
      for (int c = 0; c < 16; ++c) s[c] += (double)red[(j * G + threadIdx.x) * 16 + c];
    }
    float* p = part + (((int64_t)b * gridDim.x + blockIdx.x) * C + threadIdx.x * 8) * 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) { p[c * 2] = (float)s[c]; p[c * 2 + 1] = (float)s[8 + c]; }
  }
}

template <int DYM>
__global__ __launch_bounds__(BLK) void in_bwd_apply_bf16_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ s1, const float* __restrict__ s2,
                                                                void* __restrict__ dx, int64_t V, int C, int64_t total8,
                                                                const PoolSrc ps = PoolSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0}) {
  constexpr bool DY_BF = DYM == 1, POOL = DYM == 2;
  const int G = C >> 3;
  const bool small = total8 < (1ll << 31) && V < (1ll << 31);
  const int64_t stride = (int64_t)gridDim.x * BLK;
  auto pool_of = [&](int64_t i) {
    int g, b;
    in_elem(i, G, V, small, g, b);
    return pool_dy8_load(ps, b, i / G - (int64_t)b * V, V, C, g);
  };
  auto oneg = [&](int64_t i, const Raw8& rx, const float (&gs)[8]) {
    int g, b;
    in_elem(i, G, V, small, g, b);
    const int bc0 = b * C + g * 8;
    float xs[8], o[8], m[8], rs[8], a1[8], a2[8];
    unpack8<true>(rx, xs);
    ld8f(mean + bc0, m); ld8f(rstd + bc0, rs); ld8f(s1 + bc0, a1); ld8f(s2 + bc0, a2);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float r = rs[c];
      const float xh = (xs[c] - m[c]) * r;
      const float gg = gs[c] * (xh > 0.f ? 1.f : LRELU_SLOPE);
      o[c] = r * (gg - a1[c] - xh * a2[c]);
    }
    store8<true>(dx, i, o);
  };
  auto one = [&](int64_t i, const Raw8& rx, const Raw8& rg) {
    float gs[8];
    unpack8<DY_BF>(rg, gs);
    oneg(i, rx, gs);
  };
  constexpr int ILP = 2;
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  for (; i + (ILP - 1) * stride < total8; i += ILP * stride) {
    Raw8 rx[ILP], rg[ILP];
    PoolDy8 pd[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      rx[u] = load8_raw<true>(x, i + u * stride);
      if constexpr (POOL) pd[u] = pool_of(i + u * stride); else rg[u] = load8_raw<DY_BF>(dy, i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      raw8_fence<true>(rx[u]);
      if constexpr (POOL) pool_dy8_fence(pd[u]); else raw8_fence<DY_BF>(rg[u]);
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      if constexpr (POOL) { float gs[8]; pool_dy8_unpack(pd[u], gs); oneg(i + u * stride, rx[u], gs); }
      else one(i + u * stride, rx[u], rg[u]);
    }
  }
  for (; i < total8; i += stride) {
    if constexpr (POOL) { float gs[8]; pool_dy8_unpack(pool_of(i), gs); oneg(i, load8_raw<true>(x, i), gs); }
    else one(i, load8_raw<true>(x, i), load8_raw<DY_BF>(dy, i));
  }
}

template <bool TO_BF>
__global__ __launch_bounds__(BLK) void cast8_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t total8) {
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total8; i += (int64_t)gridDim.x * BLK) {
    float f[8];
    load8<!TO_BF>(x, i, f);
    store8<TO_BF>(y, i, f);
  }
}

}  // namespace

extern "C" {

size_t modet_instnorm_ws_bytes(int B, int64_t V, int C) {
  const int64_t nchunk = cdiv64(V, in_chunk(C));
  return ((size_t)B * nchunk * C * 2 + (size_t)B * C * 2) * sizeof(float);
}

int modet_instnorm_lrelu_fwd(const float* x, float* y, float* mean, float* rstd, void* ws, size_t ws_bytes, int B,
                             int64_t V, int C, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 4 != 0 || C > 512) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_instnorm_ws_bytes(B, V, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int chunk = in_chunk(C);
  const int nchunk = (int)cdiv64(V, chunk);
  float* part = (float*)ws;
  hipLaunchKernelGGL(in_partial_kernel<0>, dim3(nchunk, B), dim3(BLK), 0, s, x, nullptr, nullptr, nullptr, part, V, C, chunk);
  hipLaunchKernelGGL(in_finalize_kernel<0>, dim3(C, B), dim3(64), 0, s, part, mean, rstd, V, C, nchunk, eps);
  const int64_t total4 = (int64_t)B * V * (C / 4);
  hipLaunchKernelGGL(in_apply_kernel, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, s, x, y, mean, rstd, V, C, total4);
  return modet_launch_status();
}

static int rows_from_bytes(size_t stats_bytes, int B, int C, int64_t* rows) {
  // B*C shift header + rows * B*C*2 partial sums (modet_conv3d_stats_bytes)
  const int64_t n = (int64_t)(stats_bytes / sizeof(float)), bc = (int64_t)B * C;
  *rows = (n / bc - 1) / 2;
  return (*rows > 0 && (size_t)(bc + *rows * bc * 2) * sizeof(float) == stats_bytes) ? MODET_OK : MODET_ERR_DIM;
}

int modet_instnorm_lrelu_fwd_stats(const float* x, float* y, float* mean, float* rstd, const float* stats,
                                   size_t stats_bytes, int B, int64_t V, int C, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(stats);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 4 != 0 || 2 * C > 256) return MODET_ERR_UNSUPPORTED;
  int64_t rows;
  if (rows_from_bytes(stats_bytes, B, C, &rows) != MODET_OK) return MODET_ERR_DIM;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(in_rows_finalize_kernel, dim3(B, (2 * C + IN_FIN_COLS - 1) / IN_FIN_COLS), dim3(256), 0, s, stats, stats + (size_t)B * C, mean, rstd, V, C, rows, eps);
  const int64_t total4 = (int64_t)B * V * (C / 4);
  hipLaunchKernelGGL(in_apply_kernel, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, s, x, y, mean, rstd, V, C, total4);
  return modet_launch_status();
}

/* mean / rstd only (no apply pass): from the conv epilogue's partials if `stats` is given, else by a statistics pass
 * over x (ws as for modet_instnorm_lrelu_fwd) */
int modet_instnorm_stats(const float* x, float* mean, float* rstd, const float* stats, size_t stats_bytes, void* ws,
                         size_t ws_bytes, int B, int64_t V, int C, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 4 != 0 || C > 512) return MODET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (stats) {
    if (2 * C > 256) return MODET_ERR_UNSUPPORTED;
    int64_t rows;
    if (rows_from_bytes(stats_bytes, B, C, &rows) != MODET_OK) return MODET_ERR_DIM;
    hipLaunchKernelGGL(in_rows_finalize_kernel, dim3(B, (2 * C + IN_FIN_COLS - 1) / IN_FIN_COLS), dim3(256), 0, s, (const float*)stats, (const float*)stats + (size_t)B * C,
                       mean, rstd, V, C, rows, eps);
    return modet_launch_status();
  }
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(ws);
  if (ws_bytes < modet_instnorm_ws_bytes(B, V, C)) return MODET_ERR_WORKSPACE;
  const int chunk = in_chunk(C);
  const int nchunk = (int)cdiv64(V, chunk);
  float* part = (float*)ws;
  hipLaunchKernelGGL(in_partial_kernel<0>, dim3(nchunk, B), dim3(BLK), 0, s, x, nullptr, nullptr, nullptr, part, V, C, chunk);
  hipLaunchKernelGGL(in_finalize_kernel<0>, dim3(C, B), dim3(64), 0, s, part, mean, rstd, V, C, nchunk, eps);
  return modet_launch_status();
}

int modet_instnorm_lrelu_bwd(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                             void* ws, size_t ws_bytes, int B, int64_t V, int C, modet_stream_t stream) {
  return modet_instnorm_lrelu_bwd_amax(d_y, x, mean, rstd, d_x, ws, ws_bytes, B, V, C, nullptr, stream);
}

int modet_instnorm_lrelu_bwd_amax(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                                  void* ws, size_t ws_bytes, int B, int64_t V, int C, float* amax, modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(x); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(d_x);
  MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 4 != 0 || C > 512) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_instnorm_ws_bytes(B, V, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int chunk = in_chunk(C);
  const int nchunk = (int)cdiv64(V, chunk);
  float* part = (float*)ws;
  float* s1 = part + (size_t)B * nchunk * C * 2;
  float* s2 = s1 + (size_t)B * C;
  hipLaunchKernelGGL(in_partial_kernel<1>, dim3(nchunk, B), dim3(BLK), 0, s, x, d_y, mean, rstd, part, V, C, chunk);
  hipLaunchKernelGGL(in_finalize_kernel<1>, dim3(C, B), dim3(64), 0, s, part, s1, s2, V, C, nchunk, 0.f, amax);
  const int64_t total4 = (int64_t)B * V * (C / 4);
  hipLaunchKernelGGL(in_bwd_apply_kernel<false>, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, s, d_y, x, mean, rstd, s1, s2,
                     d_x, V, C, total4, PoolSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0}, amax);
  return modet_launch_status();
}

/* InstanceNorm backward whose statistics pass already happened in the producer of d_y (modet_conv3d_bwd_data_instats):
 * rows [B][rows_per_b][C][2] of (sum g, sum g*xhat) -> fixed-order fp64 column sums / V, then the apply pass.
 * ws >= 2 * B * C floats. */
int modet_instnorm_lrelu_bwd_rows(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                                  const float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B, int64_t V, int C,
                                  modet_stream_t stream) {
  return modet_instnorm_lrelu_bwd_rows_amax(d_y, x, mean, rstd, d_x, rows, rows_bytes, ws, ws_bytes, B, V, C, nullptr, stream);
}

int modet_instnorm_lrelu_bwd_rows_amax(const float* d_y, const float* x, const float* mean, const float* rstd, float* d_x,
                                       const float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B, int64_t V, int C,
                                       float* amax, modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(x); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(d_x);
  MODET_CHECK_PTR(rows); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 4 != 0 || 2 * C > 256) return MODET_ERR_UNSUPPORTED;
  const int64_t per = (int64_t)(rows_bytes / sizeof(float)) / ((int64_t)B * C * 2);
  if (per < 1 || (size_t)per * B * C * 2 * sizeof(float) != rows_bytes) return MODET_ERR_DIM;
  if (ws_bytes < (size_t)2 * B * C * sizeof(float)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* s1 = (float*)ws;
  float* s2 = s1 + (size_t)B * C;
  hipLaunchKernelGGL(in_rows_finalize_bwd_kernel, dim3(B, (2 * C + IN_FIN_COLS - 1) / IN_FIN_COLS), dim3(256), 0, s, rows, s1, s2, V, C, per, amax);
  const int64_t total4 = (int64_t)B * V * (C / 4);
  hipLaunchKernelGGL(in_bwd_apply_kernel<false>, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, s, d_y, x, mean, rstd, (const float*)s1,
                     (const float*)s2, d_x, V, C, total4, PoolSrc{nullptr, nullptr, nullptr, 0, 0, 0, 0}, amax);
  return modet_launch_status();
}

/* InstanceNorm backward of a level's output block whose gradient is unpool(g_pooled) / 8 + [add_a ; add_b] (the block's
 * output went to AvgPool3d(2) and, split into its two batch halves, to the level's consumers): the sum is formed while the two
 * passes read it instead of being written by modet_avgpool2_bwd and read back twice.  Same kernels, same order of the sums. */
int modet_instnorm_lrelu_bwd_pool(const float* g_pooled, const float* add_a, const float* add_b, int Bh, const float* x,
                                  const float* mean, const float* rstd, float* d_x, void* ws, size_t ws_bytes, int B, int D,
                                  int H, int W, int C, modet_stream_t stream) {
  return modet_instnorm_lrelu_bwd_pool_amax(g_pooled, add_a, add_b, Bh, x, mean, rstd, d_x, ws, ws_bytes, B, D, H, W, C, nullptr, stream);
}

int modet_instnorm_lrelu_bwd_pool_amax(const float* g_pooled, const float* add_a, const float* add_b, int Bh, const float* x,
                                       const float* mean, const float* rstd, float* d_x, void* ws, size_t ws_bytes, int B, int D,
                                       int H, int W, int C, float* amax, modet_stream_t stream) {
  MODET_CHECK_PTR(g_pooled); MODET_CHECK_PTR(x); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0 && Bh >= 0 && Bh <= B);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  const int64_t V = (int64_t)D * H * W;
  if (C % 4 != 0 || C > 512 || V >= (1ll << 31)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_instnorm_ws_bytes(B, V, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const PoolSrc ps{g_pooled, add_a, add_b, Bh, D, H, W};
  const int chunk = in_chunk(C);
  const int nchunk = (int)cdiv64(V, chunk);
  float* part = (float*)ws;
  float* s1 = part + (size_t)B * nchunk * C * 2;
  float* s2 = s1 + (size_t)B * C;
  hipLaunchKernelGGL(in_partial_kernel<2>, dim3(nchunk, B), dim3(BLK), 0, s, x, nullptr, mean, rstd, part, V, C, chunk, ps);
  hipLaunchKernelGGL(in_finalize_kernel<1>, dim3(C, B), dim3(64), 0, s, part, s1, s2, V, C, nchunk, 0.f, amax);
  const int64_t total4 = (int64_t)B * V * (C / 4);
  hipLaunchKernelGGL(in_bwd_apply_kernel<true>, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, s, nullptr, x, mean, rstd, s1, s2,
                     d_x, V, C, total4, ps, amax);
  return modet_launch_status();
}

int modet_lrelu_bwd(const float* d_y, const float* y, float* d_x, int64_t n, modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(y); MODET_CHECK_PTR(d_x);
  MODET_CHECK_DIM(n > 0);
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(flat_grid(n, BLK)), dim3(BLK), 0, (hipStream_t)stream, d_y, y, d_x, n);
  return modet_launch_status();
}

int modet_scale_by_dev_scalar(const float* x, const float* s, float* y, int64_t n, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(s); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(n > 0);
  hipLaunchKernelGGL(scale_kernel, dim3(flat_grid(n, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, s, y, n);
  return modet_launch_status();
}

int modet_avgpool2_fwd(const float* x, float* y, int B, int D, int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 4 != 0) return MODET_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_fwd_kernel<false>, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, y, D, H,
                     W, C, total4);
  return modet_launch_status();
}

/* AvgPool3d(2) of a bf16 tensor into an fp32 one (BASELINE.json configs[4]: a level's bf16 features -> the next level's input) */
int modet_avgpool2_fwd_x16(const void* x_bf16, float* y, int B, int D, int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(x_bf16); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 4 != 0) return MODET_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_fwd_kernel<true>, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, (hipStream_t)stream, (const float*)x_bf16,
                     y, D, H, W, C, total4);
  return modet_launch_status();
}

int modet_instnorm_lrelu_apply_pool(const float* x, const float* mean, const float* rstd, float* y, float* pooled, int B,
                                    int D, int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(y); MODET_CHECK_PTR(pooled);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 4 != 0) return MODET_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(in_apply_pool_kernel, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, y, pooled, mean,
                     rstd, D, H, W, C, total4);
  return modet_launch_status();
}

int modet_avgpool2_bwd(const float* d_y, const float* addend, float* d_x, int B, int D, int H, int W, int C,
                       modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(d_x);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 4 != 0) return MODET_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)B * D * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(flat_grid(total4, BLK)), dim3(BLK), 0, (hipStream_t)stream, d_y, addend,
                     d_x, D, H, W, C, total4);
  return modet_launch_status();
}

int modet_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr,
                            float beta1, float beta2, float eps, int step, float grad_scale, modet_stream_t stream) {
  MODET_CHECK_PTR(p); MODET_CHECK_PTR(g); MODET_CHECK_PTR(m); MODET_CHECK_PTR(v); MODET_CHECK_PTR(vmax);
  MODET_CHECK_DIM(n > 0 && step >= 1);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(n, BLK)), dim3(BLK), 0, (hipStream_t)stream, p, g, m, v, vmax, n,
                     step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
  return modet_launch_status();
}

// ---- bf16 storage (cfg 5): x is always the bf16 raw conv output, statistics fp32
size_t modet_instnorm_bf16_ws_bytes(int B, int64_t V, int C) {
  const int64_t nchunk = cdiv64(V, in_chunk8(C));
  return ((size_t)B * nchunk * C * 2 + (size_t)B * C * 2) * sizeof(float);
}

int modet_instnorm_lrelu_fwd_stats_bf16(const void* x, void* y, int y_bf16, float* mean, float* rstd, const float* stats,
                                        size_t stats_bytes, int B, int64_t V, int C, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(stats);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 8 != 0 || 2 * C > 256) return MODET_ERR_UNSUPPORTED;
  int64_t rows;
  if (rows_from_bytes(stats_bytes, B, C, &rows) != MODET_OK) return MODET_ERR_DIM;
  hipStream_t s = (hipStream_t)stream;
  // layout (modet_conv3d_bf16_stats_bytes): [B][C] shift | [B][tiles][C][2] | tail [B][IN_SLICES][C][2]
  const int64_t tiles = rows - IN_SLICES;
  if (tiles <= 0) return MODET_ERR_DIM;
  const float* trows = stats + (size_t)B * C;
  float* tail = const_cast<float*>(trows) + (size_t)B * tiles * 2 * C;
  hipLaunchKernelGGL(in_rows_slice_kernel, dim3(IN_SLICES, B), dim3(256), 0, s, trows, tail, C, tiles);
  hipLaunchKernelGGL(in_rows_finalize_kernel, dim3(B, (2 * C + IN_FIN_COLS - 1) / IN_FIN_COLS), dim3(256), 0, s, stats, (const float*)tail, mean, rstd, V, C,
                     (int64_t)IN_SLICES, eps);
  const int64_t total8 = (int64_t)B * V * (C / 8);
  if (y_bf16) hipLaunchKernelGGL(in_apply_bf16_kernel<true>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, s, x, y, mean, rstd, V, C, total8);
  else hipLaunchKernelGGL(in_apply_bf16_kernel<false>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, s, x, y, mean, rstd, V, C, total8);
  return modet_launch_status();
}

/* modet_instnorm_lrelu_fwd_stats_bf16 with y stored as bf16, plus AvgPool3d(2) of y -- of its fp32 values, before the rounding --
 * into `pooled` (fp32), in the same pass */
int modet_instnorm_lrelu_fwd_stats_pool_bf16(const void* x, void* y_bf16, float* pooled, float* mean, float* rstd, const float* stats,
                                             size_t stats_bytes, int B, int D, int H, int W, int C, float eps, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y_bf16); MODET_CHECK_PTR(pooled); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(stats);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 8 != 0 || 2 * C > 256) return MODET_ERR_UNSUPPORTED;
  const int64_t V = (int64_t)D * H * W;
  int64_t rows;
  if (rows_from_bytes(stats_bytes, B, C, &rows) != MODET_OK) return MODET_ERR_DIM;
  hipStream_t s = (hipStream_t)stream;
  const int64_t tiles = rows - IN_SLICES;
  if (tiles <= 0) return MODET_ERR_DIM;
  const float* trows = stats + (size_t)B * C;
  float* tail = const_cast<float*>(trows) + (size_t)B * tiles * 2 * C;
  hipLaunchKernelGGL(in_rows_slice_kernel, dim3(IN_SLICES, B), dim3(256), 0, s, trows, tail, C, tiles);
  hipLaunchKernelGGL(in_rows_finalize_kernel, dim3(B, (2 * C + IN_FIN_COLS - 1) / IN_FIN_COLS), dim3(256), 0, s, stats, (const float*)tail, mean, rstd, V, C,
                     (int64_t)IN_SLICES, eps);
  const int64_t total8 = (int64_t)B * (D / 2) * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(in_apply_pool_bf16_kernel, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, s, x, y_bf16, pooled, mean, rstd, D, H, W, C,
                     total8);
  return modet_launch_status();
}

int modet_instnorm_lrelu_bwd_bf16(const void* d_y, int dy_bf16, const void* x, const float* mean, const float* rstd, void* d_x,
                                  void* ws, size_t ws_bytes, int B, int64_t V, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(x); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(d_x);
  MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && V > 0 && C > 0);
  if (C % 8 != 0 || C > 512) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_instnorm_bf16_ws_bytes(B, V, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int chunk = in_chunk8(C);
  const int nchunk = (int)cdiv64(V, chunk);
  float* part = (float*)ws;
  float* s1 = part + (size_t)B * nchunk * C * 2;
  float* s2 = s1 + (size_t)B * C;
  const int64_t total8 = (int64_t)B * V * (C / 8);
  if (dy_bf16) hipLaunchKernelGGL(in_partial_bf16_kernel<1>, dim3(nchunk, B), dim3(BLK), 0, s, x, d_y, mean, rstd, part, V, C, chunk);
  else hipLaunchKernelGGL(in_partial_bf16_kernel<0>, dim3(nchunk, B), dim3(BLK), 0, s, x, d_y, mean, rstd, part, V, C, chunk);
  hipLaunchKernelGGL(in_finalize_kernel<1>, dim3(C, B), dim3(64), 0, s, part, s1, s2, V, C, nchunk, 0.f);
  if (dy_bf16) hipLaunchKernelGGL(in_bwd_apply_bf16_kernel<1>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, s, d_y, x, mean, rstd, s1, s2, d_x, V, C, total8);
  else hipLaunchKernelGGL(in_bwd_apply_bf16_kernel<0>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, s, d_y, x, mean, rstd, s1, s2, d_x, V, C, total8);
  return modet_launch_status();
}

/* modet_instnorm_lrelu_bwd_pool for the bf16 chain: x (the block's raw conv output) and d_x are bf16, the gradients of the
 * level's consumers and the pooled gradient fp32 */
int modet_instnorm_lrelu_bwd_pool_bf16(const float* g_pooled, const float* add_a, const float* add_b, int Bh, const void* x,
                                       const float* mean, const float* rstd, void* d_x, void* ws, size_t ws_bytes, int B, int D,
                                       int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(g_pooled); MODET_CHECK_PTR(x); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1 && C > 0 && Bh >= 0 && Bh <= B);
  MODET_CHECK_DIM(D % 2 == 0 && H % 2 == 0 && W % 2 == 0);
  const int64_t V = (int64_t)D * H * W;
  if (C % 8 != 0 || C > 512 || V >= (1ll << 31)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_instnorm_bf16_ws_bytes(B, V, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const PoolSrc ps{g_pooled, add_a, add_b, Bh, D, H, W};
  const int chunk = in_chunk8(C);
  const int nchunk = (int)cdiv64(V, chunk);
  float* part = (float*)ws;
  float* s1 = part + (size_t)B * nchunk * C * 2;
  float* s2 = s1 + (size_t)B * C;
  const int64_t total8 = (int64_t)B * V * (C / 8);
  hipLaunchKernelGGL(in_partial_bf16_kernel<2>, dim3(nchunk, B), dim3(BLK), 0, s, x, nullptr, mean, rstd, part, V, C, chunk, ps);
  hipLaunchKernelGGL(in_finalize_kernel<1>, dim3(C, B), dim3(64), 0, s, part, s1, s2, V, C, nchunk, 0.f);
  hipLaunchKernelGGL(in_bwd_apply_bf16_kernel<2>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, s, nullptr, x, mean, rstd, s1, s2, d_x,
                     V, C, total8, ps);
  return modet_launch_status();
}

/* y = cast(x), n % 8 == 0: to_bf16 != 0: fp32 -> bf16 (round to nearest even), else bf16 -> fp32 */
int modet_cast_bf16(const void* x, void* y, int64_t n, int to_bf16, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(n > 0 && n % 8 == 0);
  const int64_t total8 = n / 8;
  if (to_bf16) hipLaunchKernelGGL(cast8_kernel<true>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, y, total8);
  else hipLaunchKernelGGL(cast8_kernel<false>, dim3(flat_grid(total8, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, y, total8);
  return modet_launch_status();
}

}  // extern "C"
