// Shared helpers for the gfx950 kernels of libmodet_hip.so (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/modet_hip.h"

#define MODET_CHECK_PTR(p) do { if ((p) == nullptr) return MODET_ERR_NULL; } while (0)
#define MODET_CHECK_DIM(c) do { if (!(c)) return MODET_ERR_DIM; } while (0)

// every entry point ends with this: report launch-time errors to the caller
static inline int modet_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MODET_OK : (int)e;
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
