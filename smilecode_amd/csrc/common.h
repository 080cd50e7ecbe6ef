// Shared helpers for the gfx950 kernels of libmodet_hip.so (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/modet_hip.h"

#define MODET_CHECK_PTR(p) do { if ((p) == nullptr) return MODET_ERR_NULL; } while (0)
#define MODET_CHECK_DIM(c) do { if (!(c)) return MODET_ERR_DIM; } while (0)

// every entry point ends with this: report launch-time errors to the caller
static inline int modet_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MODET_OK : (int)e;
}

// A/B switches of the kernel dispatch exist in TUNING builds only (-DMODET_TUNING: `MODET_TUNING=1 python -m smilecode_amd.build`
// or tools/build_variant.sh): the product library reads no environment variable, so its behaviour -- which kernel family a
// shape runs, hence its rounding -- is a function of the arguments alone.  Returns the variable's first character, or 0.
static inline char modet_tuning_env(const char* name) {
#ifdef MODET_TUNING
  const char* e = getenv(name);
  return e ? e[0] : 0;
#else
  (void)name;
  return 0;
#endif
}

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// grid for a flat memory-bound kernel: at most 256 CUs x 8 blocks, grid-stride the rest
static inline int flat_grid(int64_t n, int block) {
  int64_t g = cdiv64(n, block);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// Zero fill as a KERNEL.  hipMemsetAsync must not be used by anything that may be captured into a hipGraph: on this ROCm
// (7.2) a captured memset node clears its buffer on the first replay only -- later replays leave garbage (tools/
// exp_graph_memset.py: warp backward captured alone, replay 0 exact, replay 1 off by 2e27).  n_bytes % 4 == 0.
static __global__ void modet_zero_kernel(unsigned* __restrict__ p, int64_t n_words) {
  const int64_t n4 = n_words >> 2;
  uint4* p4 = reinterpret_cast<uint4*>(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    p4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[(n4 << 2) + threadIdx.x] = 0u;
}
static inline void modet_zero_async(void* p, size_t n_bytes, hipStream_t s) {     // p 16-byte aligned (torch allocations are)
  const int64_t words = (int64_t)(n_bytes / 4);
  hipLaunchKernelGGL(modet_zero_kernel, dim3(flat_grid(words / 4 + 1, 256)), dim3(256), 0, s, (unsigned*)p, words);
}

#define LRELU_SLOPE 0.1f

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : LRELU_SLOPE * v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of one float per thread (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}

// Column sums of a row-major [rows][ncol] float matrix over the rows [r0, r1), by one 256-thread workgroup, in fp64:
// thread t reads element t of each group of 256/ncol consecutive rows (fully coalesced), four independent
// accumulators, then a fixed-order sum over the row lanes.  out[ncol] is written by threads < ncol.  ncol <= 256.
// Fixed assignment + fixed order: deterministic.  Used as stage 1 of the two-stage statistics reductions
// (stage 2 adds the per-slice results in slice order).
constexpr int COLSUM_SLICES = 64;
// `stride` > 0: the matrix has `stride` columns per row and this call sums its columns [base .. base + ncol) only (a column
// group per workgroup: the per-column order of the additions does not depend on the grouping's width only on `per`).
template <typename T>
__device__ __forceinline__ void block_colsum_256(const T* __restrict__ base, int64_t r0, int64_t r1, int ncol,
                                                 double* __restrict__ out, double* sm /* [256] shared */, int stride = 0) {
  const int per = 256 / ncol;
  const int col = threadIdx.x % ncol, rl = threadIdx.x / ncol;
  const int64_t rs = stride > 0 ? stride : ncol;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (rl < per) {
    int64_t r = r0 + rl;
    for (; r + 3 * per < r1; r += 4 * per) {
      a0 += (double)base[r * rs + col];
      a1 += (double)base[(r + per) * rs + col];
      a2 += (double)base[(r + 2 * per) * rs + col];
      a3 += (double)base[(r + 3 * per) * rs + col];
    }
    for (; r < r1; r += per) a0 += (double)base[r * rs + col];
  }
  sm[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < ncol) {
    double t = 0.0;
    for (int k = 0; k < per; ++k) t += sm[k * ncol + threadIdx.x];
    out[threadIdx.x] = t;
  }
}

