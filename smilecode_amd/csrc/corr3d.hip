// 27-displacement local correlation of the PR++ baseline ("Baseline methods/PR++/models.py":205-232, Correlation3D with
// kernel_size 3, d = 3, sw = 1, sf = 2) -- SURVEY.md 8(f) rank 4, a sibling of the neighbourhood attention:
//   pm = box3(mov), pf = box3(fix)  (3x3x3 all-ones grouped conv, zero padded; pf also on a 1-voxel ring outside the
//   volume, where the box still overlaps it -- the reference pads fix by sf+1 = 3),
//   corr[b][t][p] = (1/27) sum_c pm[b,p,c] * pf[b, p + 2*off(t), c],   t = 9i+3j+k, off = (i-1, j-1, k-1).
// Channels-last features (B,D,H,W,C), C % 4 == 0; corr is written as (B,27,D,H,W) like the reference.
// Backward: d_pm and d_pf by the transposed gathers, then the box sum (self-adjoint under zero padding) once more.
// No atomics; HBM/L2-bound gathers.
#include "common.h"

namespace {

constexpr int BLK = 256;

// out[(b, z, y, x), c4] = sum over the 27 neighbours of in, both on grids extended by e_out / e_in voxels per side
// (values outside an input grid are zero).  Output voxel q (extended coords) has volume position q - e_out.
__global__ __launch_bounds__(BLK) void box3_kernel(const float* __restrict__ in, float* __restrict__ out, int D, int H,
                                                   int W, int C, int e_in, int e_out, unsigned total) {
  const unsigned G = C >> 2;
  const int Do = D + 2 * e_out, Ho = H + 2 * e_out, Wo = W + 2 * e_out;
  const int Di = D + 2 * e_in, Hi = H + 2 * e_in, Wi = W + 2 * e_in;
  for (unsigned idx = blockIdx.x * BLK + threadIdx.x; idx < total; idx += gridDim.x * BLK) {
    const unsigned g = idx % G;
    unsigned r = idx / G;
    const int x = (int)(r % Wo) - e_out; r /= Wo;
    const int y = (int)(r % Ho) - e_out; r /= Ho;
    const int z = (int)(r % Do) - e_out;
    const int b = (int)(r / Do);
    // the 27 loads are unconditional (clamped coordinates; the box outside the padded volume is selected to zero afterwards) and
    // fenced in rows of nine: under `if (inside) continue` each one compiled to its own memory round trip, 27 in series
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
      const int zz = z + dz + e_in;
      const bool zk = zz >= 0 && zz < Di;
      float4 v[9];
      bool ok[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int yy = y + q / 3 - 1 + e_in, xx = x + q % 3 - 1 + e_in;
        ok[q] = zk && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
        const int64_t off = ok[q] ? ((((int64_t)b * Di + zz) * Hi + yy) * Wi + xx) * C : (int64_t)b * Di * Hi * Wi * C;
        v[q] = *reinterpret_cast<const float4*>(in + off + g * 4);
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
#pragma unroll
      for (int q = 0; q < 9; ++q)
        if (ok[q]) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    *reinterpret_cast<float4*>(out + (int64_t)idx * 4) = acc;
  }
}

// corr[b][t][p] = (1/27) pm[p] . pfx[p + 2 off(t)]; pfx lives on the grid extended by 1 (centres at -2 / dim+1 are zero).
// One workgroup per strip of 32 consecutive voxels: GL = C/4 lanes share one (voxel, displacement) dot product (each a
// float4 of both operands: contiguous reads), partial sums meet by xor-shuffles (GL a power of two) or a short serial
// loop, the 27 x 32 results go through LDS so that every correlation plane receives one contiguous 128 B row.
constexpr int CS_V = 32;
__global__ __launch_bounds__(BLK) void corr_fwd_kernel(const float* __restrict__ pm, const float* __restrict__ pfx,
                                                       float* __restrict__ corr, int D, int H, int W, int C, int64_t N) {
  __shared__ float res[27 * CS_V];
  const int GL = C >> 2;
  const bool pow2 = (GL & (GL - 1)) == 0 && GL <= 64;
  const int He = H + 2, We = W + 2;
  const int64_t V = (int64_t)D * H * W;
  const int64_t n0 = (int64_t)blockIdx.x * CS_V;
  const int items = 27 * CS_V * (pow2 ? GL : 1);
  for (int it = threadIdx.x; it < items; it += BLK) {
    const int j = pow2 ? it % GL : 0;
    const int pair = pow2 ? it / GL : it;                // = v * 27 + t
    const int t = pair % 27, v = pair / 27;
    const int64_t n = n0 + v;
    float s = 0.f;
    if (n < N) {
      const int64_t b = n / V, p = n - b * V;
      const int x = (int)(p % W);
      const int64_t r = p / W;
      const int y = (int)(r % H), z = (int)(r / H);
      const int qz = z + 2 * (t / 9 - 1), qy = y + 2 * ((t / 3) % 3 - 1), qx = x + 2 * (t % 3 - 1);
      if (qz >= -1 && qz <= D && qy >= -1 && qy <= H && qx >= -1 && qx <= W) {
        const float* a = pm + n * C;
        const float* f = pfx + (((b * (D + 2) + qz + 1) * He + qy + 1) * We + qx + 1) * C;
        if (pow2) {
          const float4 av = *reinterpret_cast<const float4*>(a + j * 4), fv = *reinterpret_cast<const float4*>(f + j * 4);
          s = av.x * fv.x + av.y * fv.y + av.z * fv.z + av.w * fv.w;
        } else {
          for (int c = 0; c < C; c += 4) {
            const float4 av = *reinterpret_cast<const float4*>(a + c), fv = *reinterpret_cast<const float4*>(f + c);
            s = fmaf(av.x, fv.x, s); s = fmaf(av.y, fv.y, s); s = fmaf(av.z, fv.z, s); s = fmaf(av.w, fv.w, s);
          }
        }
      }
    }
    if (pow2) {
      for (int o = 1; o < GL; o <<= 1) s += __shfl_xor(s, o, 64);     // the GL lanes of a pair are adjacent (BLK % GL == 0)
      if (j == 0) res[t * CS_V + v] = s;
    } else {
      res[t * CS_V + v] = s;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 27 * CS_V; i += BLK) {
    const int t = i / CS_V, v = i - t * CS_V;
    const int64_t n = n0 + v;
    if (n < N) {
      const int64_t b = n / V, p = n - b * V;
      corr[(b * 27 + t) * V + p] = res[i] * (1.f / 27.f);
    }
  }
}

// d_pm[p][c4] = (1/27) sum_t g[t][p] pfx[p + 2 off(t)][c4]
__global__ __launch_bounds__(BLK) void corr_bwd_pm_kernel(const float* __restrict__ g, const float* __restrict__ pfx,
                                                          float* __restrict__ dpm, int D, int H, int W, int C,
                                                          unsigned total) {
  const unsigned G = C >> 2;
  const int He = H + 2, We = W + 2;
  const int64_t V = (int64_t)D * H * W;
  for (unsigned idx = blockIdx.x * BLK + threadIdx.x; idx < total; idx += gridDim.x * BLK) {
    const unsigned c4 = idx % G;
    unsigned r = idx / G;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int z = (int)(r % D);
    const int b = (int)(r / D);
    const int64_t p = ((int64_t)z * H + y) * W + x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // nine displacements per batch, loads unconditional (clamped) and fenced together; the sum keeps its order
#pragma unroll
    for (int t0 = 0; t0 < 27; t0 += 9) {
      float gv[9];
      float4 fv[9];
      bool ok[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int t = t0 + u;
        const int qz = z + 2 * (t / 9 - 1), qy = y + 2 * ((t / 3) % 3 - 1), qx = x + 2 * (t % 3 - 1);
        ok[u] = qz >= -1 && qz <= D && qy >= -1 && qy <= H && qx >= -1 && qx <= W;
        gv[u] = g[((int64_t)b * 27 + t) * V + p];
        const int64_t off = ok[u] ? ((((int64_t)b * (D + 2) + qz + 1) * He + qy + 1) * We + qx + 1) * C : (int64_t)b * (D + 2) * He * We * C;
        fv[u] = *reinterpret_cast<const float4*>(pfx + off + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) asm volatile("" : "+v"(gv[u]), "+v"(fv[u].x), "+v"(fv[u].y), "+v"(fv[u].z), "+v"(fv[u].w));
#pragma unroll
      for (int u = 0; u < 9; ++u)
        if (ok[u]) {
          acc.x = fmaf(gv[u], fv[u].x, acc.x); acc.y = fmaf(gv[u], fv[u].y, acc.y);
          acc.z = fmaf(gv[u], fv[u].z, acc.z); acc.w = fmaf(gv[u], fv[u].w, acc.w);
        }
    }
    const float k = 1.f / 27.f;
    *reinterpret_cast<float4*>(dpm + (int64_t)idx * 4) = make_float4(acc.x * k, acc.y * k, acc.z * k, acc.w * k);
  }
}

// d_pfx[q][c4] = (1/27) sum_t g[t][q - 2 off(t)] pm[q - 2 off(t)][c4]   (q on the extended grid)
__global__ __launch_bounds__(BLK) void corr_bwd_pf_kernel(const float* __restrict__ g, const float* __restrict__ pm,
                                                          float* __restrict__ dpfx, int D, int H, int W, int C,
                                                          unsigned total) {
  const unsigned G = C >> 2;
  const int De = D + 2, He = H + 2, We = W + 2;
  const int64_t V = (int64_t)D * H * W;
  for (unsigned idx = blockIdx.x * BLK + threadIdx.x; idx < total; idx += gridDim.x * BLK) {
    const unsigned c4 = idx % G;
    unsigned r = idx / G;
    const int qx = (int)(r % We) - 1; r /= We;
    const int qy = (int)(r % He) - 1; r /= He;
    const int qz = (int)(r % De) - 1;
    const int b = (int)(r / De);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t0 = 0; t0 < 27; t0 += 9) {                  // as in corr_bwd_pm_kernel: nine fenced (g, pm) pairs per batch
      float gv[9];
      float4 av[9];
      bool ok[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int t = t0 + u;
        const int z = qz - 2 * (t / 9 - 1), y = qy - 2 * ((t / 3) % 3 - 1), x = qx - 2 * (t % 3 - 1);
        ok[u] = z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
        const int64_t p = ok[u] ? ((int64_t)z * H + y) * W + x : 0;
        gv[u] = g[((int64_t)b * 27 + t) * V + p];
        av[u] = *reinterpret_cast<const float4*>(pm + ((int64_t)b * V + p) * C + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) asm volatile("" : "+v"(gv[u]), "+v"(av[u].x), "+v"(av[u].y), "+v"(av[u].z), "+v"(av[u].w));
#pragma unroll
      for (int u = 0; u < 9; ++u)
        if (ok[u]) {
          acc.x = fmaf(gv[u], av[u].x, acc.x); acc.y = fmaf(gv[u], av[u].y, acc.y);
          acc.z = fmaf(gv[u], av[u].z, acc.z); acc.w = fmaf(gv[u], av[u].w, acc.w);
        }
    }
    const float k = 1.f / 27.f;
    *reinterpret_cast<float4*>(dpfx + (int64_t)idx * 4) = make_float4(acc.x * k, acc.y * k, acc.z * k, acc.w * k);
  }
}

inline int64_t ext_elems(int B, int D, int H, int W, int C) { return (int64_t)B * (D + 2) * (H + 2) * (W + 2) * C; }

}  // namespace

extern "C" {

size_t modet_corr3d_ws_bytes(int B, int D, int H, int W, int C) {
  // pm | pfx | d_pm | d_pfx  (the forward uses the first two)
  return (size_t)(2 * ((int64_t)B * D * H * W * C + ext_elems(B, D, H, W, C))) * sizeof(float);
}

static int corr_check(int B, int D, int H, int W, int C) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MODET_ERR_DIM;
  if (C % 4 != 0) return MODET_ERR_UNSUPPORTED;
  if (ext_elems(B, D, H, W, C) >= ((int64_t)1 << 31) || (int64_t)B * D * H * W * 27 >= ((int64_t)1 << 31)) return MODET_ERR_UNSUPPORTED;
  return MODET_OK;
}

int modet_corr3d_fwd(const float* mov, const float* fix, float* corr, void* ws, size_t ws_bytes, int B, int D, int H,
                     int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(mov); MODET_CHECK_PTR(fix); MODET_CHECK_PTR(corr); MODET_CHECK_PTR(ws);
  if (const int e = corr_check(B, D, H, W, C)) return e;
  if (ws_bytes < modet_corr3d_ws_bytes(B, D, H, W, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)B * D * H * W, ne = ext_elems(B, D, H, W, C) / C;
  float* pm = (float*)ws;
  float* pfx = pm + n * C;
  hipLaunchKernelGGL(box3_kernel, dim3(flat_grid(n * (C / 4), BLK)), dim3(BLK), 0, s, mov, pm, D, H, W, C, 0, 0, (unsigned)(n * (C / 4)));
  hipLaunchKernelGGL(box3_kernel, dim3(flat_grid(ne * (C / 4), BLK)), dim3(BLK), 0, s, fix, pfx, D, H, W, C, 0, 1, (unsigned)(ne * (C / 4)));
  hipLaunchKernelGGL(corr_fwd_kernel, dim3((unsigned)cdiv64(n, CS_V)), dim3(BLK), 0, s, (const float*)pm, (const float*)pfx, corr, D, H,
                     W, C, n);
  return modet_launch_status();
}

int modet_corr3d_bwd(const float* mov, const float* fix, const float* d_corr, float* d_mov, float* d_fix, void* ws,
                     size_t ws_bytes, int B, int D, int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(mov); MODET_CHECK_PTR(fix); MODET_CHECK_PTR(d_corr); MODET_CHECK_PTR(d_mov); MODET_CHECK_PTR(d_fix);
  MODET_CHECK_PTR(ws);
  if (const int e = corr_check(B, D, H, W, C)) return e;
  if (ws_bytes < modet_corr3d_ws_bytes(B, D, H, W, C)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)B * D * H * W, ne = ext_elems(B, D, H, W, C) / C;
  const unsigned G = C / 4;
  float* pm = (float*)ws;
  float* pfx = pm + n * C;
  float* dpm = pfx + ne * C;
  float* dpfx = dpm + n * C;
  hipLaunchKernelGGL(box3_kernel, dim3(flat_grid(n * G, BLK)), dim3(BLK), 0, s, mov, pm, D, H, W, C, 0, 0, (unsigned)(n * G));
  hipLaunchKernelGGL(box3_kernel, dim3(flat_grid(ne * G, BLK)), dim3(BLK), 0, s, fix, pfx, D, H, W, C, 0, 1, (unsigned)(ne * G));
  hipLaunchKernelGGL(corr_bwd_pm_kernel, dim3(flat_grid(n * G, BLK)), dim3(BLK), 0, s, d_corr, (const float*)pfx, dpm, D, H, W, C,
                     (unsigned)(n * G));
  hipLaunchKernelGGL(corr_bwd_pf_kernel, dim3(flat_grid(ne * G, BLK)), dim3(BLK), 0, s, d_corr, (const float*)pm, dpfx, D, H, W, C,
                     (unsigned)(ne * G));
  // the box sum is self-adjoint: d_mov = box3(d_pm); d_fix = box3 of d_pfx (extended grid) evaluated inside the volume
  hipLaunchKernelGGL(box3_kernel, dim3(flat_grid(n * G, BLK)), dim3(BLK), 0, s, (const float*)dpm, d_mov, D, H, W, C, 0, 0, (unsigned)(n * G));
  hipLaunchKernelGGL(box3_kernel, dim3(flat_grid(n * G, BLK)), dim3(BLK), 0, s, (const float*)dpfx, d_fix, D, H, W, C, 1, 0, (unsigned)(n * G));
  return modet_launch_status();
}

}  // extern "C"
