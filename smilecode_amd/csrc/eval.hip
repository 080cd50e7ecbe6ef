// Evaluation tail, part 2: fraction of voxels with a non-positive Jacobian determinant of the deformation
// (reference ModeT/utils.py:108-150 jacobian_determinant_vxm + ModeT/infer.py:89-90 `np.sum(jac_det <= 0)`).
//
// The reference does this on the host in float64: flow.cpu().numpy() (a 59 MB D2H per pair) + the int64 identity grid,
// np.gradient (central differences, one-sided at the ends), the 3x3 determinant expanded along its first row.  Here one
// kernel reads the flow where the model left it and returns ONE integer per sample.  The count is integer-exact against
// the reference because every fp64 operation is performed in the same order with the same (IEEE, round-to-nearest)
// roundings -- hence no FMA contraction in this file.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int BLK = 256;

// f_a(z,y,x) = float64(flow[z,y,x,a]) + float64(index along axis a)   (utils.py:126-130: float32 + int64 -> float64)
__device__ __forceinline__ double fval(const float* __restrict__ fl, int64_t vox, int a, int idx) {
  return (double)fl[vox * 3 + a] + (double)idx;
}

// np.gradient along one axis at position i of n (unit spacing): (f[i+1] - f[i-1]) / 2 inside, f[1] - f[0] and
// f[n-1] - f[n-2] at the ends (numpy divides those by the spacing 1.0: a no-op)
__device__ __forceinline__ void grad3(const float* __restrict__ fl, int64_t vox, int64_t stride, int i, int n, int axis,
                                      int z, int y, int x, double (&g)[3]) {
  const int lo = i > 0 ? i - 1 : i, hi = i + 1 < n ? i + 1 : i;
  const int64_t vlo = vox + (int64_t)(lo - i) * stride, vhi = vox + (int64_t)(hi - i) * stride;
  const bool inner = i > 0 && i + 1 < n;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // the identity grid's component a at the two sample points: only axis == a moves along this difference
    const int c = a == 0 ? z : (a == 1 ? y : x);
    const int clo = axis == a ? lo : c, chi = axis == a ? hi : c;
    const double d = fval(fl, vhi, a, chi) - fval(fl, vlo, a, clo);
    g[a] = inner ? d / 2.0 : d;
  }
}

// flow (B,D,H,W,3) channels-last fp32.  counts[b] += #voxels with det <= 0; det_out (B,D,H,W) fp64 optional.
__global__ __launch_bounds__(BLK) void jacdet_kernel(const float* __restrict__ flow, unsigned long long* __restrict__ counts,
                                                     double* __restrict__ det_out, int D, int H, int W, int64_t V) {
  const int b = blockIdx.y;
  const float* fl = flow + (int64_t)b * V * 3;
  unsigned int mine = 0;
  for (int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x; v < V; v += (int64_t)gridDim.x * BLK) {
    const int x = (int)(v % W);
    const int64_t t = v / W;
    const int y = (int)(t % H), z = (int)(t / H);
    double dx[3], dy[3], dz[3];                 // the reference's names: J[0], J[1], J[2] = d/d(axis 0,1,2) (utils.py:134-136)
    grad3(fl, v, (int64_t)H * W, z, D, 0, z, y, x, dx);
    grad3(fl, v, (int64_t)W, y, H, 1, z, y, x, dy);
    grad3(fl, v, 1, x, W, 2, z, y, x, dz);
    const double j0 = dx[0] * (dy[1] * dz[2] - dy[2] * dz[1]);      // utils.py:139-141
    const double j1 = dx[1] * (dy[0] * dz[2] - dy[2] * dz[0]);
    const double j2 = dx[2] * (dy[0] * dz[1] - dy[1] * dz[0]);
    const double det = j0 - j1 + j2;                                // utils.py:144
    if (det_out) det_out[(int64_t)b * V + v] = det;
    mine += det <= 0.0 ? 1u : 0u;
  }
  // integer reduction: wave shuffle sum, one atomic per wave (exact in any order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&counts[b], (unsigned long long)mine);
}

}  // namespace

extern "C" {

int modet_jacdet_nonpos_count(const float* flow, int64_t* counts, double* det_out, int B, int D, int H, int W,
                              modet_stream_t stream) {
  MODET_CHECK_PTR(flow); MODET_CHECK_PTR(counts);
  MODET_CHECK_DIM(B > 0 && D > 1 && H > 1 && W > 1);          // np.gradient needs at least 2 samples per axis
  hipStream_t s = (hipStream_t)stream;
  modet_zero_async(counts, (size_t)B * sizeof(int64_t), s);
  const int64_t V = (int64_t)D * H * W;
  int grid = flat_grid(V, BLK);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(jacdet_kernel, dim3(grid, B), dim3(BLK), 0, s, flow, (unsigned long long*)counts, det_out, D, H, W, V);
  return modet_launch_status();
}

}  // extern "C"
