// Two-stage deterministic fp64 reduction of per-workgroup d_rpb partial rows (shared by na.hip and qk_op.hip).
#pragma once
#include "common.h"

namespace {

// partial (B, heads, nblk, 27) -> out (heads,27), two deterministic fp64 stages:
//   1: grid (COLSUM_SLICES, heads, B): coalesced column sums of a slice of the nblk rows -> scratch[b][h][slice][27]
//   2: one workgroup: (b, slice) added in order per (h, t)
template <typename T>
__global__ __launch_bounds__(256) void drpb_stage1_kernel(const T* __restrict__ part, double* __restrict__ scratch,
                                                          int heads, int64_t nblk) {
  __shared__ double sm[256];
  const int sl = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int64_t per = cdiv64(nblk, COLSUM_SLICES);
  const int64_t r0 = sl * per, r1 = r0 + per < nblk ? r0 + per : nblk;
  const int64_t bh = (int64_t)b * heads + h;
  block_colsum_256(part + bh * nblk * 27, r0 < r1 ? r0 : r1, r1, 27, scratch + (bh * COLSUM_SLICES + sl) * 27, sm);
}
template <typename T>
__global__ __launch_bounds__(256) void drpb_stage2_kernel(const double* __restrict__ scratch, T* __restrict__ out,
                                                          int B, int heads) {
  for (int i = threadIdx.x; i < heads * 27; i += 256) {
    const int h = i / 27, t = i - h * 27;
    double s = 0.0;
    for (int b = 0; b < B; ++b)
      for (int sl = 0; sl < COLSUM_SLICES; ++sl) s += scratch[(((int64_t)b * heads + h) * COLSUM_SLICES + sl) * 27 + t];
    out[i] = (T)s;
  }
}
inline size_t drpb_scratch_bytes(int B, int heads) { return (size_t)B * heads * COLSUM_SLICES * 27 * sizeof(double); }
// `part` rows start at ws; the scratch sits at byte offset `scratch_off` (8-byte aligned) of the same workspace
template <typename T>
inline void drpb_reduce(const T* part, void* ws, size_t scratch_off, T* d_rpb, int B, int heads, int64_t nblk, hipStream_t s) {
  double* scr = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + scratch_off);
  hipLaunchKernelGGL(drpb_stage1_kernel<T>, dim3(COLSUM_SLICES, heads, B), dim3(256), 0, s, part, scr, heads, nblk);
  hipLaunchKernelGGL(drpb_stage2_kernel<T>, dim3(1), dim3(256), 0, s, (const double*)scr, d_rpb, B, heads);
}

}  // namespace
