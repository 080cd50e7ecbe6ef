// Neighbourhood (3x3x3) attention of the motion-decomposition head, gfx950.
//
//  * modet_na_fwd / modet_na_bwd : fused ModeTransformer.forward (reference ModeT/models.py:308-334,
//    ModeT-cu/models.py:300-316): logits -> softmax over the 27 modes -> expected offset, one thread per
//    (voxel, head) owning all 27 logits in registers; the K tile (+1-voxel halo) is staged in LDS once
//    per workgroup and read 27x from there.  The (..,27) attention tensor is never written to HBM.
//  * modet_qk_fwd / modet_qk_bwd : the reference CUDA operator's exact tensor contract
//    (ModeT-cu/modet/modet_kernel.cu:17-381), for callers shaped like ModeT-cu/functional.py.
#include "common.h"
#include "drpb_reduce.h"
#include <cstdlib>

namespace {

constexpr int HD = 6;          // head_dim (train.py:49); the fused kernels are specialised for it
constexpr int TZ = 4, TY = 4, TX = 16;           // voxel tile of one workgroup (256 threads)
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int HVOX = HZ * HY * HX;               // 648 halo'd voxels
constexpr int NTHREADS = TZ * TY * TX;

struct TileGeom {
  int tiles_x, tiles_y, tiles_z;
};

__device__ __forceinline__ void tile_origin(int tile, const TileGeom g, int& z0, int& y0, int& x0) {
  const int tx = tile % g.tiles_x;
  const int t2 = tile / g.tiles_x;
  x0 = tx * TX;
  y0 = (t2 % g.tiles_y) * TY;
  z0 = (t2 / g.tiles_y) * TZ;
}

// stage k[b, z0-1.., y0-1.., x0-1.., head*6 + 0..5] into LDS, zeros outside the volume.  Two phases: ALL global loads of a
// thread are issued first (clamped, always valid addresses; the halo outside the volume is selected to zero afterwards),
// then the LDS writes.  As one loop with a load per `if (inside)` every item was its own memory round trip: eight in
// series per tile (and nineteen in na_bwd_kernel) in front of ~10 us of work.
constexpr int KT_IT = (HVOX * 3 + NTHREADS - 1) / NTHREADS;
// QK16 (cfg 5, bf16 storage of q / k): the tensor holds bf16, two channels per 32-bit word; everything past the load is fp32
__device__ __forceinline__ float2 bf2_unpack(unsigned u) { return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)); }
template <bool QK16 = false>
__device__ __forceinline__ void stage_k_load(float2 (&r)[KT_IT], const float* __restrict__ k, int64_t bbase, int z0, int y0,
                                             int x0, int D, int H, int W, int C, int hoff) {
  bool inside[KT_IT];
#pragma unroll
  for (int j = 0; j < KT_IT; ++j) {
    const int idx = min((int)threadIdx.x + j * NTHREADS, HVOX * 3 - 1);
    const int v = idx / 3, part = idx - v * 3;
    const int hx = v % HX, t = v / HX;
    const int hy = t % HY, hz = t / HY;
    const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
    const bool in = z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
    const int64_t off = (bbase + (in ? ((int64_t)z * H + y) * W + x : 0)) * C + hoff + part * 2;
    if constexpr (QK16) r[j] = bf2_unpack(reinterpret_cast<const unsigned*>(k)[off >> 1]);      // (C, hoff even: off is even)
    else r[j] = *reinterpret_cast<const float2*>(k + off);
    inside[j] = in;
  }
  // the loaded values are "used" here, unconditionally and all at once: without this hipcc sinks every load back under its
  // `inside` test (branch, load, s_waitcnt vmcnt(0), eight times in series)
#pragma unroll
  for (int j = 0; j < KT_IT; ++j) asm volatile("" : "+v"(r[j].x), "+v"(r[j].y));
#pragma unroll
  for (int j = 0; j < KT_IT; ++j)
    if (!inside[j]) r[j] = make_float2(0.f, 0.f);
}
__device__ __forceinline__ void stage_k_store(float* __restrict__ kt, const float2 (&r)[KT_IT]) {
#pragma unroll
  for (int j = 0; j < KT_IT; ++j) {
    const int idx = threadIdx.x + j * NTHREADS;          // = v * 3 + part, and kt + v * HD + part * 2 = kt + idx * 2
    if (idx < HVOX * 3) *reinterpret_cast<float2*>(kt + idx * 2) = r[j];
  }
}
template <bool QK16 = false>
__device__ __forceinline__ void stage_k_tile(float* __restrict__ kt, const float* __restrict__ k, int64_t bbase,
                                             int z0, int y0, int x0, int D, int H, int W, int C, int hoff) {
  float2 r[KT_IT];
  stage_k_load<QK16>(r, k, bbase, z0, y0, x0, D, H, W, C, hoff);
  stage_k_store(kt, r);
}

__device__ __forceinline__ void load6(const float* __restrict__ p, float (&r)[HD]) {
  const float2 a = *reinterpret_cast<const float2*>(p);
  const float2 b = *reinterpret_cast<const float2*>(p + 2);
  const float2 c = *reinterpret_cast<const float2*>(p + 4);
  r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y; r[4] = c.x; r[5] = c.y;
}

template <bool QK16>
__device__ __forceinline__ void load6g(const float* __restrict__ base, int64_t el, float (&r)[HD]) {     // el: element index (even)
  if constexpr (QK16) {
    const unsigned* p = reinterpret_cast<const unsigned*>(base) + (el >> 1);
    const float2 a = bf2_unpack(p[0]), b = bf2_unpack(p[1]), c = bf2_unpack(p[2]);
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y; r[4] = c.x; r[5] = c.y;
  } else {
    load6(base + el, r);
  }
}

// 27 logits of one voxel from the LDS tile; lt = linear halo index of the voxel's (-1,-1,-1) neighbour
__device__ __forceinline__ void logits27(const float* __restrict__ kt, int lt, const float (&qs)[HD],
                                         const float* __restrict__ rpb_h, float (&lg)[27]) {
#pragma unroll
  for (int ki = 0; ki < 3; ++ki)
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int t = ki * 9 + kj * 3 + kk;
        float kv[HD];
        load6(kt + (lt + (ki * HY + kj) * HX + kk) * HD, kv);
        float s = rpb_h[t];
#pragma unroll
        for (int c = 0; c < HD; ++c) s = fmaf(qs[c], kv[c], s);
        lg[t] = s;
      }
}

__device__ __forceinline__ float softmax27(float (&lg)[27], float& m) {   // lg -> exp(lg - max); returns 1/sum
  m = lg[0];
#pragma unroll
  for (int t = 1; t < 27; ++t) m = fmaxf(m, lg[t]);
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    lg[t] = __expf(lg[t] - m);
    s += lg[t];
  }
  return 1.f / s;
}

// ------------------------------------------------------------------------------------------ fused forward
template <bool QK16>
__global__ __launch_bounds__(NTHREADS) void na_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ rpb, float* __restrict__ out,
                                                          float* __restrict__ lse, int D, int H, int W, int heads,
                                                          float scale, TileGeom g) {
  __shared__ __attribute__((aligned(16))) float kt[HVOX * HD];
  __shared__ float rp[27];
  const int h = blockIdx.y, b = blockIdx.z;
  const int C = heads * HD;
  const int64_t V = (int64_t)D * H * W;
  int z0, y0, x0;
  tile_origin(blockIdx.x, g, z0, y0, x0);
  if (threadIdx.x < 27) rp[threadIdx.x] = rpb[h * 27 + threadIdx.x];
  stage_k_tile<QK16>(kt, k, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * HD);
  __syncthreads();

  const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tz = threadIdx.x / (TX * TY);
  const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
  if (z >= D || y >= H || x >= W) return;
  const int64_t n = (int64_t)b * V + ((int64_t)z * H + y) * W + x;
  float qs[HD];
  load6g<QK16>(q, n * C + h * HD, qs);
#pragma unroll
  for (int c = 0; c < HD; ++c) qs[c] *= scale;
  float p[27];
  logits27(kt, (tz * HY + ty) * HX + tx, qs, rp, p);
  float mx;
  const float inv = softmax27(p, mx);
  if (lse) lse[n * heads + h] = mx - __logf(inv);      // log-sum-exp of the 27 logits, for the single-pass backward
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
  for (int ki = 0; ki < 3; ++ki)
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const float pv = p[ki * 9 + kj * 3 + kk];
        o0 += pv * (float)(ki - 1);
        o1 += pv * (float)(kj - 1);
        o2 += pv * (float)(kk - 1);
      }
  float* o = out + n * (heads * 3) + h * 3;
  o[0] = o0 * inv; o[1] = o1 * inv; o[2] = o2 * inv;
}

// Sum 27 per-lane values over the 64 lanes of a wave: butterfly reduce-scatter over a 32-slot array (slots 27..31
// zero).  At step M the lanes with bit M set keep the upper half of the live slots and hand the lower half to their
// partner (and vice versa), halving the live slots; after M = 32,16,8,4,2 one slot is left, summed over 32 lanes, and
// the last exchange (M = 1) completes it.  Lane l returns the wave total of slot (l >> 1) (bits 5..1 select it).
template <int N>     // one butterfly step: N live slots -> N/2, partner = lane ^ N
__device__ __forceinline__ void rs_step(float (&v)[32], int lane) {
  const bool up = (lane & N) != 0;
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const float keep = up ? v[i + N / 2] : v[i];
    const float send = up ? v[i] : v[i + N / 2];
    v[i] = keep + __shfl_xor(send, N, 64);
  }
}
__device__ __forceinline__ float wave_reduce_scatter32(const float (&v27)[27], int lane) {
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = i < 27 ? v27[i] : 0.f;
  // live slots 32 -> 16 -> 8 -> 4 -> 2 -> 1 with partners lane ^ 32, 16, 8, 4, 2 (slot count == xor mask here)
  rs_step<32>(v, lane); rs_step<16>(v, lane); rs_step<8>(v, lane); rs_step<4>(v, lane); rs_step<2>(v, lane);
  return v[0] + __shfl_xor(v[0], 1, 64);
}

// ------------------------------------------------------------------------------------------ backward (one pass)
// With lse[n] = logsumexp_t logit[n][t] saved by the forward and u[n] = d_out[n].out[n] (out = E[offset]),
//   dlogit[n][t] = p[n][t] * (d_out[n].off(t) - u[n]),   p[n][t] = exp(logit[n][t] - lse[n]).
// A thread owns voxel n in two roles:
//   query role: its 27 logits against the K tile -> d_q[n] = scale * sum_t dlogit[n][t] k[n+off(t)], d_rpb partial;
//   key role  : for each t the query voxel m = n - off(t) has logit[m][t] = scale q[m].k[n] + rpb[t] (k[n] is its own
//               key), so p[m][t] needs only q[m], lse[m]: d_k[n] = scale * sum_t dlogit[m][t] q[m].
// q (pre-scaled), k, d_out, u and lse tiles with a 1-voxel halo live in LDS; nothing 27-wide ever goes to HBM (the
// previous two-pass version wrote and re-read a 27-plane dlogit scratch: 216 B of its 324 B per voxel-head).
// No atomics, deterministic.
constexpr int AUX = 5;      // per halo voxel: d_out[3], u, lse

template <bool QK16>
__global__ __launch_bounds__(NTHREADS) void na_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ rpb, const float* __restrict__ out,
                                                          const float* __restrict__ lse, const float* __restrict__ dout,
                                                          float* __restrict__ dq, float* __restrict__ dk,
                                                          float* __restrict__ drpb_part, int D, int H, int W, int heads,
                                                          float scale, TileGeom g) {
  __shared__ __attribute__((aligned(16))) float kt[HVOX * HD];
  __shared__ __attribute__((aligned(16))) float qt[HVOX * HD];
  __shared__ float ax[HVOX * AUX];
  __shared__ float red[27 * (NTHREADS / 64)];
  __shared__ float rp[27];
  const int h = blockIdx.y, b = blockIdx.z;
  const int C = heads * HD;
  const int64_t V = (int64_t)D * H * W;
  int z0, y0, x0;
  tile_origin(blockIdx.x, g, z0, y0, x0);
  if (threadIdx.x < 27) rp[threadIdx.x] = rpb[h * 27 + threadIdx.x];
#ifdef NA_BWD_LDS_PAD            // tools: resource-footprint variants for the preemption hunt (never defined in the product build)
  __shared__ float lds_pad[NA_BWD_LDS_PAD];
  if (g.tiles_x < 0) { lds_pad[threadIdx.x % NA_BWD_LDS_PAD] = scale; __syncthreads(); rp[0] += lds_pad[(threadIdx.x * 7 + 1) % NA_BWD_LDS_PAD]; }
#endif
#ifdef NA_BWD_VGPR_CLOBBER
#define NA_STR2(x) #x
#define NA_STR(x) NA_STR2(x)
  asm volatile("" ::: "v" NA_STR(NA_BWD_VGPR_CLOBBER));
#endif
  {
    // every global load of the three staged tiles is in flight before the first LDS write (see stage_k_load)
    constexpr int AX_IT = (HVOX + NTHREADS - 1) / NTHREADS;
    float2 rk[KT_IT], rq[KT_IT];
    float ra[AX_IT][7];
    bool ain[AX_IT];
    stage_k_load<QK16>(rk, k, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * HD);
    stage_k_load<QK16>(rq, q, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * HD);
#pragma unroll
    for (int j = 0; j < AX_IT; ++j) {
      const int v = min((int)threadIdx.x + j * NTHREADS, HVOX - 1);
      const int hx = v % HX, t = v / HX;
      const int hy = t % HY, hz = t / HY;
      const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
      const bool in = z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
      const int64_t n = (int64_t)b * V + (in ? ((int64_t)z * H + y) * W + x : 0);
      const float* go = dout + n * (heads * 3) + h * 3;
      const float* oo = out + n * (heads * 3) + h * 3;
      ra[j][0] = go[0]; ra[j][1] = go[1]; ra[j][2] = go[2];
      ra[j][3] = oo[0]; ra[j][4] = oo[1]; ra[j][5] = oo[2];
      ra[j][6] = lse[n * heads + h];
      ain[j] = in;
    }
#pragma unroll
    for (int j = 0; j < AX_IT; ++j)
#pragma unroll
      for (int e = 0; e < 7; ++e) asm volatile("" : "+v"(ra[j][e]));
#pragma unroll
    for (int j = 0; j < AX_IT; ++j)
      if (!ain[j]) {
#pragma unroll
        for (int e = 0; e < 7; ++e) ra[j][e] = 0.f;
      }
    stage_k_store(kt, rk);
    stage_k_store(qt, rq);
#pragma unroll
    for (int j = 0; j < AX_IT; ++j) {
      const int v = threadIdx.x + j * NTHREADS;
      if (v < HVOX) {
        float* a = ax + v * AUX;
        a[0] = ra[j][0]; a[1] = ra[j][1]; a[2] = ra[j][2];
        a[3] = ra[j][0] * ra[j][3] + ra[j][1] * ra[j][4] + ra[j][2] * ra[j][5];
        a[4] = ra[j][6];
      }
    }
  }
  __syncthreads();
  const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tz = threadIdx.x / (TX * TY);
  const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
  const bool live = (z < D && y < H && x < W);
  float dlv[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) dlv[t] = 0.f;
  if (live) {
    const int64_t n = (int64_t)b * V + ((int64_t)z * H + y) * W + x;
    const int lt = (tz * HY + ty) * HX + tx;             // halo index of the (-1,-1,-1) neighbour
    const int lc = lt + (HY + 1) * HX + 1;               // halo index of the voxel itself
    // ---- query role
    float qs[HD];
    load6(qt + lc * HD, qs);
#pragma unroll
    for (int c = 0; c < HD; ++c) qs[c] *= scale;
    logits27(kt, lt, qs, rp, dlv);
    const float* ac = ax + lc * AUX;
    const float g0 = ac[0], g1 = ac[1], g2 = ac[2], u = ac[3], l = ac[4];
    float dqa[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dqa[c] = 0.f;
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const int t = ki * 9 + kj * 3 + kk;
          const float gt = (float)(ki - 1) * g0 + (float)(kj - 1) * g1 + (float)(kk - 1) * g2;
          const float d = __expf(dlv[t] - l) * (gt - u);
          dlv[t] = d;
          float kv[HD];
          load6(kt + (lt + (ki * HY + kj) * HX + kk) * HD, kv);
#pragma unroll
          for (int c = 0; c < HD; ++c) dqa[c] = fmaf(d, kv[c], dqa[c]);
        }
    float* dqp = dq + n * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 2) *reinterpret_cast<float2*>(dqp + c) = make_float2(dqa[c] * scale, dqa[c + 1] * scale);
  }
  // wave sums of the 27 dlogits as one butterfly reduce-scatter (32 shuffles instead of 27 x 6): lane l ends up
  // with the wave total of tap slot(l); fixed pattern -> deterministic.  Done here so the 27 registers are free
  // during the key role.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    const float r = wave_reduce_scatter32(dlv, lane);
    if ((lane & 1) == 0 && (lane >> 1) < 27) red[wv * 27 + (lane >> 1)] = r;
  }
  if (live) {
    const int64_t n = (int64_t)b * V + ((int64_t)z * H + y) * W + x;
    const int lc = ((tz + 1) * HY + ty + 1) * HX + tx + 1;
    // ---- key role
    float kn[HD];
    load6(kt + lc * HD, kn);
#pragma unroll
    for (int c = 0; c < HD; ++c) kn[c] *= scale;
    float dka[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dka[c] = 0.f;
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          // query voxel m = n - off(t).  No bounds test: a halo voxel outside the volume was staged with
          // d_out = u = 0, so its dlogit is p * (0 - 0) = 0 (p stays finite: logit = rpb[t], lse = 0)
          const int lm = lc + ((1 - ki) * HY + (1 - kj)) * HX + (1 - kk);
          float qv[HD];
          load6(qt + lm * HD, qv);
          float lg = rp[ki * 9 + kj * 3 + kk];
#pragma unroll
          for (int c = 0; c < HD; ++c) lg = fmaf(qv[c], kn[c], lg);
          const float* am = ax + lm * AUX;
          const float gt = (float)(ki - 1) * am[0] + (float)(kj - 1) * am[1] + (float)(kk - 1) * am[2];
          const float d = __expf(lg - am[4]) * (gt - am[3]);
#pragma unroll
          for (int c = 0; c < HD; ++c) dka[c] = fmaf(d, qv[c], dka[c]);
        }
    float* dkp = dk + n * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 2) *reinterpret_cast<float2*>(dkp + c) = make_float2(dka[c] * scale, dka[c + 1] * scale);
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    float t = 0.f;
    for (int w = 0; w < NTHREADS / 64; ++w) t += red[w * 27 + threadIdx.x];
    const int64_t blk = ((int64_t)b * gridDim.y + h) * gridDim.x + blockIdx.x;
    drpb_part[blk * 27 + threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------ backward, z-marching
// The same single-pass arithmetic as na_bwd_kernel for the LARGE levels (pyramid levels 1-2: 0.6-4.9 M voxels, where a
// 4x4x16 tile stages 2.5x its own voxels -- 13 scalar / 8-byte loads per halo voxel -- and nothing overlaps the staging).
// A workgroup owns an 8 x 32 column of (y, x), one thread per voxel, and MARCHES along z over a chunk of planes: the
// (k, scale*q, d_out, u, lse) records of a plane (+1 halo in y and x: 1.33x) are loaded ONCE, through per-plane buffer
// descriptors (out-of-volume voxels read zeros: no bounds tests, no selects), into registers while the previous plane is
// being computed, then written into a 3-slot LDS ring; plane z needs slots z-1, z, z+1 in both roles.  The d_rpb partial
// sums stay in registers for the whole chunk: one row per workgroup.
constexpr int MY = 8, MX = 32, MHY = MY + 2, MHX = MX + 2, MHV = MHY * MHX;       // 340 halo'd voxels per plane
constexpr int MSLOT = MHV * (2 * HD + AUX);                                       // floats per ring slot
using NaBuf = __amdgpu_buffer_rsrc_t;
constexpr unsigned NA_OOB = 0x80000000u;
__device__ __forceinline__ NaBuf na_rsrc(const float* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
struct NaMarchArgs {
  const float* q; const float* k; const float* rpb; const float* out; const float* lse; const float* dout;
  float* dq; float* dk; float* drpb_part;
  int D, H, W, heads, tiles_x, tiles_y, nchunk, ZC;
  float scale;
};

template <bool QK16>
__global__ __launch_bounds__(NTHREADS, 2) void na_bwd_march_kernel(const NaMarchArgs a) {
  // ring slot: k[MHV][6] | q'[MHV][6] | (g0, g1, g2, u)[MHV] | lse'[MHV]   with q' = scale * log2(e) * q, lse' = log2(e) * lse:
  // the logits are kept in the base-2 domain (p = exp2(logit' - lse'): no multiply in front of v_exp_f32), d_k divides the
  // factor out again at the end
  __shared__ __attribute__((aligned(16))) float ring[3 * MSLOT];
  __shared__ float red[27 * (NTHREADS / 64)];
  constexpr float LOG2E = 1.4426950408889634f;
  const int h = blockIdx.y, b = blockIdx.z;
  const int D = a.D, H = a.H, W = a.W, heads = a.heads;
  const int C = heads * HD, C3 = heads * 3;
  int t = blockIdx.x;
  const int x0 = (t % a.tiles_x) * MX; t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * MY;
  const int zs = (t / a.tiles_y) * a.ZC;
  const int ze = zs + a.ZC < D ? zs + a.ZC : D;
  // the 27 biases of this head: wave-uniform, read through the scalar unit (they stay in SGPRs, no LDS traffic)
  float rp[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) rp[i] = a.rpb[h * 27 + i];  // (unscaled: the log2(e) rides on the fma that subtracts lse')
  const int64_t V = (int64_t)D * H * W, HW = (int64_t)H * W;
  constexpr int QSZ = QK16 ? 2 : 4;                         // bytes per q / k element in HBM
  const float* qb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.q) + ((int64_t)b * V * C + h * HD) * QSZ);
  const float* kb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.k) + ((int64_t)b * V * C + h * HD) * QSZ);
  const float* gb = a.dout + (int64_t)b * V * C3 + h * 3;
  const float* ob = a.out + (int64_t)b * V * C3 + h * 3;
  const float* lb = a.lse + (int64_t)b * V * heads + h;
  const unsigned qbytes = (unsigned)(HW * C * QSZ), gbytes = (unsigned)(HW * C3 * 4), lbytes = (unsigned)(HW * heads * 4);
  const float qmul = a.scale * LOG2E;

  // staging items: halo voxel v = tid (and v = 256 + tid for the first MHV - 256 threads)
  constexpr int NIT = (MHV + NTHREADS - 1) / NTHREADS;
  unsigned oq[NIT], og[NIT], ol[NIT];
  int sl_[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int v = threadIdx.x + j * NTHREADS;
    const bool on = v < MHV;
    const int hy = on ? v / MHX : 0, hx = on ? v - hy * MHX : 0;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool ok = on && y >= 0 && y < H && x >= 0 && x < W;
    const unsigned n = (unsigned)(y * W + x);
    oq[j] = ok ? n * (unsigned)(C * QSZ) : NA_OOB;
    og[j] = ok ? n * (unsigned)(C3 * 4) : NA_OOB;
    ol[j] = ok ? n * (unsigned)(heads * 4) : NA_OOB;
    sl_[j] = on ? v : -1;
  }
  float2 rk[NIT][3], rq[NIT][3];
  float rg[NIT][3], ro[NIT][3], rl[NIT];
  auto load_plane = [&](int z) {
    const bool live = z >= 0 && z < D;
    const int64_t zo = live ? z : 0;
    const NaBuf bq = na_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(qb) + zo * HW * C * QSZ), live ? qbytes : 0u);
    const NaBuf bk = na_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(kb) + zo * HW * C * QSZ), live ? qbytes : 0u);
    const NaBuf bg = na_rsrc(gb + zo * HW * C3, live ? gbytes : 0u), bo = na_rsrc(ob + zo * HW * C3, live ? gbytes : 0u);
    const NaBuf bl = na_rsrc(lb + zo * HW * heads, live ? lbytes : 0u);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if constexpr (QK16) {
          rk[j][c] = bf2_unpack(__builtin_amdgcn_raw_buffer_load_b32(bk, (int)(oq[j] + c * 4), 0, 0));
          rq[j][c] = bf2_unpack(__builtin_amdgcn_raw_buffer_load_b32(bq, (int)(oq[j] + c * 4), 0, 0));
        } else {
          rk[j][c] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(bk, (int)(oq[j] + c * 8), 0, 0));
          rq[j][c] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(bq, (int)(oq[j] + c * 8), 0, 0));
        }
        rg[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bg, (int)(og[j] + c * 4), 0, 0));
        ro[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bo, (int)(og[j] + c * 4), 0, 0));
      }
      rl[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bl, (int)ol[j], 0, 0));
    }
  };
  auto store_plane = [&](int slot) {
    float* kt = ring + slot * MSLOT;
    float* qt = kt + MHV * HD;
    float* gu = qt + MHV * HD;
    float* ls = gu + MHV * 4;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if (sl_[j] < 0) continue;
      const int v = sl_[j];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        *reinterpret_cast<float2*>(kt + v * HD + c * 2) = rk[j][c];
        *reinterpret_cast<float2*>(qt + v * HD + c * 2) = make_float2(rq[j][c].x * qmul, rq[j][c].y * qmul);
      }
      *reinterpret_cast<float4*>(gu + v * 4) =
          make_float4(rg[j][0], rg[j][1], rg[j][2], rg[j][0] * ro[j][0] + rg[j][1] * ro[j][1] + rg[j][2] * ro[j][2]);
      ls[v] = rl[j] * LOG2E;
    }
  };

  const int tx = threadIdx.x % MX, ty = threadIdx.x / MX;
  const int y = y0 + ty, x = x0 + tx;
  const bool live = y < H && x < W;
  const int hc = (ty + 1) * MHX + tx + 1;                   // this voxel inside a halo'd plane
  float dlv[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) dlv[i] = 0.f;

  // prologue: planes zs-1, zs, zs+1 -> slots 0, 1, 2; registers <- plane zs+2
  load_plane(zs - 1); store_plane(0);
  load_plane(zs); store_plane(1);
  load_plane(zs + 1); store_plane(2);
  load_plane(zs + 2);
  __syncthreads();
  for (int z = zs, i = 0; z < ze; ++z, ++i) {
    const int s0 = i % 3, s1 = (i + 1) % 3, s2 = (i + 2) % 3;          // slots of planes z-1, z, z+1
    if (live) {
      const int sof[3] = {s0 * MSLOT, s1 * MSLOT, s2 * MSLOT};
      const float* kc = ring + sof[1];
      const int64_t n = (int64_t)b * V + ((int64_t)z * H + y) * W + x;
      // ---- query role: this voxel's 27 logits against the keys around it
      float qs[HD], dqa[HD];
      load6(kc + MHV * HD + hc * HD, qs);                   // scale * log2(e) * q
#pragma unroll
      for (int c = 0; c < HD; ++c) dqa[c] = 0.f;
      const float4 gc = *reinterpret_cast<const float4*>(kc + 2 * MHV * HD + hc * 4);
      const float l = kc[2 * MHV * HD + MHV * 4 + hc];
#pragma unroll
      for (int ki = 0; ki < 3; ++ki) {
        const float* kt = ring + sof[ki];
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const int tt = ki * 9 + kj * 3 + kk;
            float kv[HD];
            load6(kt + (hc + (kj - 1) * MHX + (kk - 1)) * HD, kv);
            float lg = fmaf(rp[tt], LOG2E, -l);
#pragma unroll
            for (int c = 0; c < HD; ++c) lg = fmaf(qs[c], kv[c], lg);
            const float gt = (float)(ki - 1) * gc.x + (float)(kj - 1) * gc.y + (float)(kk - 1) * gc.z;
            const float d = __builtin_amdgcn_exp2f(lg) * (gt - gc.w);
            dlv[tt] += d;
#pragma unroll
            for (int c = 0; c < HD; ++c) dqa[c] = fmaf(d, kv[c], dqa[c]);
          }
      }
      float* dqp = a.dq + n * C + h * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 2) *reinterpret_cast<float2*>(dqp + c) = make_float2(dqa[c] * a.scale, dqa[c + 1] * a.scale);
      // ---- key role: for each tap the query voxel m = n - off(t) sees this voxel's key under that tap.  No bounds test: a
      // halo voxel outside the volume was staged with d_out = u = 0, its dlogit is p * (0 - 0) = 0 (p finite: lse = 0)
      float kn[HD], dka[HD];
      load6(kc + hc * HD, kn);
#pragma unroll
      for (int c = 0; c < HD; ++c) dka[c] = 0.f;
#pragma unroll
      for (int ki = 0; ki < 3; ++ki) {
        const float* qt = ring + sof[2 - ki] + MHV * HD;   // plane z + (1 - ki)
        const float* gu = qt + MHV * HD;
        const float* ls = gu + MHV * 4;
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const int lm = hc + (1 - kj) * MHX + (1 - kk);
            float qv[HD];
            load6(qt + lm * HD, qv);
            float lg = fmaf(rp[ki * 9 + kj * 3 + kk], LOG2E, -ls[lm]);
#pragma unroll
            for (int c = 0; c < HD; ++c) lg = fmaf(qv[c], kn[c], lg);
            const float4 gm = *reinterpret_cast<const float4*>(gu + lm * 4);
            const float gt = (float)(ki - 1) * gm.x + (float)(kj - 1) * gm.y + (float)(kk - 1) * gm.z;
            const float d = __builtin_amdgcn_exp2f(lg) * (gt - gm.w);
#pragma unroll
            for (int c = 0; c < HD; ++c) dka[c] = fmaf(d, qv[c], dka[c]);
          }
      }
      float* dkp = a.dk + n * C + h * HD;
      constexpr float ILOG2E = 0.6931471805599453f;          // q' carries log2(e): d_k = sum d * scale * q
#pragma unroll
      for (int c = 0; c < HD; c += 2) *reinterpret_cast<float2*>(dkp + c) = make_float2(dka[c] * ILOG2E, dka[c + 1] * ILOG2E);
    }
    __syncthreads();                                        // every thread is done with plane z-1 (slot s0)
    store_plane(s0);                                        // plane z+2 takes its place
    __syncthreads();
    load_plane(z + 3);                                      // in flight during the next plane's arithmetic
  }
  // d_rpb: wave sums of the 27 accumulated dlogits (butterfly reduce-scatter), then the waves through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    const float r = wave_reduce_scatter32(dlv, lane);
    if ((lane & 1) == 0 && (lane >> 1) < 27) red[wv * 27 + (lane >> 1)] = r;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    float s = 0.f;
    for (int w = 0; w < NTHREADS / 64; ++w) s += red[w * 27 + threadIdx.x];
    const int64_t blk = ((int64_t)b * gridDim.y + h) * gridDim.x + blockIdx.x;
    a.drpb_part[blk * 27 + threadIdx.x] = s;
  }
}

struct NaMarchPlan { int tiles_x, tiles_y, nchunk, zc; bool on; };
inline NaMarchPlan na_march_plan(int B, int D, int H, int W, int heads, int hd) {
  NaMarchPlan p;
  p.tiles_x = cdiv(W, MX); p.tiles_y = cdiv(H, MY);
  // the large levels only (the small ones are latency-bound either way and the 4x4x16 tiles give them more workgroups),
  // planes below 2 GiB per tensor (32-bit descriptor offsets)
  p.on = hd == HD && (int64_t)D * H * W >= 1500000 && (int64_t)H * W * heads * HD * 4 < 0x7fffffffLL;
  const int cols = B * heads * p.tiles_x * p.tiles_y;
  // z chunks of >= 8 planes, about WANT workgroups in all: the kernel is latency-bound (two waves per SIMD), many short
  // workgroups that the dispatcher interleaves beat few long ones although every chunk pays three planes of prologue
  // (measured at level 1: 1560 workgroups of 13 planes 0.37 ms, 480 of 40 planes 0.40 ms)
  int want = 2000;
#ifdef MODET_TUNING
  if (const char* e = getenv("MODET_NA_WGS")) want = atoi(e);
  if (const char* e = getenv("MODET_NA_MARCH")) p.on = p.on && e[0] != '0';
#endif
  int best = (want + cols - 1) / cols;
  const int maxn = D / 8 > 0 ? D / 8 : 1;
  best = best < 1 ? 1 : (best > maxn ? maxn : best);
  p.zc = cdiv(D, best);
  p.nchunk = cdiv(D, p.zc);
  return p;
}

// ------------------------------------------------------------------------------------------ any head_dim % 8 == 0
// The same two kernels for head dimensions other than ModeT's 6 (8 ... 128): e.g. Im2Grid's CoTr
// ("Baseline methods/Im2Grid/models.py":276-322) is this attention with one head over all C channels, no bias, no
// scale.  Channels are walked in chunks of 8: the K (and, backward, Q) halo tile of the current chunk lives in LDS and
// the 27 logits accumulate in registers.  The backward makes two passes over the chunks: logits of both roles first,
// then d_q / d_k chunk by chunk.
constexpr int GC = 8;       // channels per chunk

__device__ __forceinline__ void stage_chunk(float* __restrict__ dst, const float* __restrict__ srcp, int64_t bbase, int z0,
                                            int y0, int x0, int D, int H, int W, int C, int coff) {
  for (int idx = threadIdx.x; idx < HVOX * 2; idx += NTHREADS) {
    const int v = idx >> 1, part = idx & 1;
    const int hx = v % HX, t = v / HX;
    const int hy = t % HY, hz = t / HY;
    const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W)
      val = *reinterpret_cast<const float4*>(srcp + (bbase + ((int64_t)z * H + y) * W + x) * C + coff + part * 4);
    *reinterpret_cast<float4*>(dst + v * GC + part * 4) = val;
  }
}
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&r)[GC]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
}
__device__ __forceinline__ float dot8(const float (&a)[GC], const float (&b)[GC]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < GC; ++c) s = fmaf(a[c], b[c], s);
  return s;
}

__global__ __launch_bounds__(NTHREADS) void na_fwd_gen_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ rpb, float* __restrict__ out,
                                                              float* __restrict__ lse, int D, int H, int W, int heads,
                                                              int hd, float scale, TileGeom g) {
  __shared__ __attribute__((aligned(16))) float kt[HVOX * GC];
  __shared__ float rp[27];
  const int h = blockIdx.y, b = blockIdx.z;
  const int C = heads * hd;
  const int64_t V = (int64_t)D * H * W;
  int z0, y0, x0;
  tile_origin(blockIdx.x, g, z0, y0, x0);
  if (threadIdx.x < 27) rp[threadIdx.x] = rpb[h * 27 + threadIdx.x];
  const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tz = threadIdx.x / (TX * TY);
  const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
  const bool live = z < D && y < H && x < W;
  const int64_t n = (int64_t)b * V + ((int64_t)(live ? z : 0) * H + (live ? y : 0)) * W + (live ? x : 0);
  const int lt = (tz * HY + ty) * HX + tx;
  float p[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) p[t] = 0.f;
  for (int c0 = 0; c0 < hd; c0 += GC) {
    __syncthreads();
    stage_chunk(kt, k, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * hd + c0);
    __syncthreads();
    if (live) {
      float qs[GC];
      load8(q + n * C + h * hd + c0, qs);
#pragma unroll
      for (int ki = 0; ki < 3; ++ki)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            float kv[GC];
            load8(kt + (lt + (ki * HY + kj) * HX + kk) * GC, kv);
            p[ki * 9 + kj * 3 + kk] += dot8(qs, kv);
          }
    }
  }
  if (!live) return;
#pragma unroll
  for (int t = 0; t < 27; ++t) p[t] = fmaf(p[t], scale, rp[t]);
  float mx;
  const float inv = softmax27(p, mx);
  if (lse) lse[n * heads + h] = mx - __logf(inv);
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
  for (int ki = 0; ki < 3; ++ki)
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const float pv = p[ki * 9 + kj * 3 + kk];
        o0 += pv * (float)(ki - 1);
        o1 += pv * (float)(kj - 1);
        o2 += pv * (float)(kk - 1);
      }
  float* o = out + n * (heads * 3) + h * 3;
  o[0] = o0 * inv; o[1] = o1 * inv; o[2] = o2 * inv;
}

__global__ __launch_bounds__(NTHREADS) void na_bwd_gen_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ rpb, const float* __restrict__ out,
                                                              const float* __restrict__ lse, const float* __restrict__ dout,
                                                              float* __restrict__ dq, float* __restrict__ dk,
                                                              float* __restrict__ drpb_part, int D, int H, int W,
                                                              int heads, int hd, float scale, TileGeom g) {
  __shared__ __attribute__((aligned(16))) float kt[HVOX * GC];
  __shared__ __attribute__((aligned(16))) float qt[HVOX * GC];
  __shared__ float ax[HVOX * AUX];
  __shared__ float red[27 * (NTHREADS / 64)];
  __shared__ float rp[27];
  const int h = blockIdx.y, b = blockIdx.z;
  const int C = heads * hd;
  const int64_t V = (int64_t)D * H * W;
  int z0, y0, x0;
  tile_origin(blockIdx.x, g, z0, y0, x0);
  if (threadIdx.x < 27) rp[threadIdx.x] = rpb[h * 27 + threadIdx.x];
  for (int v = threadIdx.x; v < HVOX; v += NTHREADS) {
    const int hx = v % HX, t = v / HX;
    const int hy = t % HY, hz = t / HY;
    const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, u = 0.f, l = 0.f;
    if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) {
      const int64_t n = (int64_t)b * V + ((int64_t)z * H + y) * W + x;
      const float* go = dout + n * (heads * 3) + h * 3;
      const float* oo = out + n * (heads * 3) + h * 3;
      g0 = go[0]; g1 = go[1]; g2 = go[2];
      u = g0 * oo[0] + g1 * oo[1] + g2 * oo[2];
      l = lse[n * heads + h];
    }
    float* a = ax + v * AUX;
    a[0] = g0; a[1] = g1; a[2] = g2; a[3] = u; a[4] = l;
  }
  const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tz = threadIdx.x / (TX * TY);
  const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
  const bool live = (z < D && y < H && x < W);
  const int lt = (tz * HY + ty) * HX + tx;
  const int lc = lt + (HY + 1) * HX + 1;
  // ---- pass A: logits of both roles, accumulated over the channel chunks
  float dlq[27], dlk[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) { dlq[t] = 0.f; dlk[t] = 0.f; }
  for (int c0 = 0; c0 < hd; c0 += GC) {
    __syncthreads();
    stage_chunk(kt, k, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * hd + c0);
    stage_chunk(qt, q, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * hd + c0);
    __syncthreads();
    if (live) {
      float qn[GC], kn[GC];
      load8(qt + lc * GC, qn);
      load8(kt + lc * GC, kn);
#pragma unroll
      for (int ki = 0; ki < 3; ++ki)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const int t = ki * 9 + kj * 3 + kk;
            float v8[GC];
            load8(kt + (lt + (ki * HY + kj) * HX + kk) * GC, v8);
            dlq[t] += dot8(qn, v8);
            load8(qt + (lc + ((1 - ki) * HY + (1 - kj)) * HX + (1 - kk)) * GC, v8);
            dlk[t] += dot8(v8, kn);
          }
    }
  }
  if (live) {
    const float* ac = ax + lc * AUX;
    const float g0 = ac[0], g1 = ac[1], g2 = ac[2], u = ac[3], l = ac[4];
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const int t = ki * 9 + kj * 3 + kk;
          const float gt = (float)(ki - 1) * g0 + (float)(kj - 1) * g1 + (float)(kk - 1) * g2;
          dlq[t] = __expf(fmaf(dlq[t], scale, rp[t]) - l) * (gt - u);
          const float* am = ax + (lc + ((1 - ki) * HY + (1 - kj)) * HX + (1 - kk)) * AUX;
          const float gm = (float)(ki - 1) * am[0] + (float)(kj - 1) * am[1] + (float)(kk - 1) * am[2];
          dlk[t] = __expf(fmaf(dlk[t], scale, rp[t]) - am[4]) * (gm - am[3]);
        }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    const float r = wave_reduce_scatter32(dlq, lane);       // dlq is all zero on dead lanes
    if ((lane & 1) == 0 && (lane >> 1) < 27) red[wv * 27 + (lane >> 1)] = r;
  }
  // ---- pass B: d_q / d_k chunk by chunk
  const int64_t n = (int64_t)b * V + ((int64_t)(live ? z : 0) * H + (live ? y : 0)) * W + (live ? x : 0);
  for (int c0 = 0; c0 < hd; c0 += GC) {
    __syncthreads();
    stage_chunk(kt, k, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * hd + c0);
    stage_chunk(qt, q, (int64_t)b * V, z0, y0, x0, D, H, W, C, h * hd + c0);
    __syncthreads();
    if (live) {
      float dqa[GC], dka[GC];
#pragma unroll
      for (int c = 0; c < GC; ++c) { dqa[c] = 0.f; dka[c] = 0.f; }
#pragma unroll
      for (int ki = 0; ki < 3; ++ki)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const int t = ki * 9 + kj * 3 + kk;
            float v8[GC];
            load8(kt + (lt + (ki * HY + kj) * HX + kk) * GC, v8);
#pragma unroll
            for (int c = 0; c < GC; ++c) dqa[c] = fmaf(dlq[t], v8[c], dqa[c]);
            load8(qt + (lc + ((1 - ki) * HY + (1 - kj)) * HX + (1 - kk)) * GC, v8);
#pragma unroll
            for (int c = 0; c < GC; ++c) dka[c] = fmaf(dlk[t], v8[c], dka[c]);
          }
      float* dqp = dq + n * C + h * hd + c0;
      float* dkp = dk + n * C + h * hd + c0;
      *reinterpret_cast<float4*>(dqp) = make_float4(dqa[0] * scale, dqa[1] * scale, dqa[2] * scale, dqa[3] * scale);
      *reinterpret_cast<float4*>(dqp + 4) = make_float4(dqa[4] * scale, dqa[5] * scale, dqa[6] * scale, dqa[7] * scale);
      *reinterpret_cast<float4*>(dkp) = make_float4(dka[0] * scale, dka[1] * scale, dka[2] * scale, dka[3] * scale);
      *reinterpret_cast<float4*>(dkp + 4) = make_float4(dka[4] * scale, dka[5] * scale, dka[6] * scale, dka[7] * scale);
    }
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    float t = 0.f;
    for (int w = 0; w < NTHREADS / 64; ++w) t += red[w * 27 + threadIdx.x];
    const int64_t blk = ((int64_t)b * gridDim.y + h) * gridDim.x + blockIdx.x;
    drpb_part[blk * 27 + threadIdx.x] = t;
  }
}

inline bool gen_hd_ok(int hd) { return hd >= GC && hd <= 128 && hd % GC == 0; }

inline TileGeom geom(int D, int H, int W) { return TileGeom{cdiv(W, TX), cdiv(H, TY), cdiv(D, TZ)}; }

}  // namespace

extern "C" {

int modet_na_fwd_t(const void* q, const void* k, int qk_bf16, const float* rpb, float* out, float* lse, int B, int D, int H, int W,
                   int heads, int hd, float scale, modet_stream_t stream) {
  if (!qk_bf16) return modet_na_fwd((const float*)q, (const float*)k, rpb, out, lse, B, D, H, W, heads, hd, scale, stream);
  MODET_CHECK_PTR(q); MODET_CHECK_PTR(k); MODET_CHECK_PTR(rpb); MODET_CHECK_PTR(out);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && heads > 0);
  if (hd != HD) return MODET_ERR_UNSUPPORTED;
  const TileGeom g = geom(D, H, W);
  dim3 grid(g.tiles_x * g.tiles_y * g.tiles_z, heads, B);
  hipLaunchKernelGGL(na_fwd_kernel<true>, grid, dim3(NTHREADS), 0, (hipStream_t)stream, (const float*)q, (const float*)k, rpb, out,
                     lse, D, H, W, heads, scale, g);
  return modet_launch_status();
}

int modet_na_fwd(const float* q, const float* k, const float* rpb, float* out, float* lse, int B, int D, int H, int W,
                 int heads, int hd, float scale, modet_stream_t stream) {
  MODET_CHECK_PTR(q); MODET_CHECK_PTR(k); MODET_CHECK_PTR(rpb); MODET_CHECK_PTR(out);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && heads > 0);
  if (hd != HD && !gen_hd_ok(hd)) return MODET_ERR_UNSUPPORTED;
  const TileGeom g = geom(D, H, W);
  dim3 grid(g.tiles_x * g.tiles_y * g.tiles_z, heads, B);
  if (hd == HD)
    hipLaunchKernelGGL(na_fwd_kernel<false>, grid, dim3(NTHREADS), 0, (hipStream_t)stream, q, k, rpb, out, lse, D, H, W, heads,
                       scale, g);
  else
    hipLaunchKernelGGL(na_fwd_gen_kernel, grid, dim3(NTHREADS), 0, (hipStream_t)stream, q, k, rpb, out, lse, D, H, W,
                       heads, hd, scale, g);
  return modet_launch_status();
}

size_t modet_na_bwd_ws_bytes(int B, int D, int H, int W, int heads) {
  const TileGeom g = geom(D, H, W);                      // (at least as many rows as the z-marching kernel's workgroups)
  size_t fl = (size_t)B * heads * g.tiles_x * g.tiles_y * g.tiles_z * 27;
  fl += fl & 1;                                        // keep the fp64 scratch that follows 8-byte aligned
  return fl * sizeof(float) + drpb_scratch_bytes(B, heads);
}

int modet_na_bwd_t(const void* qv, const void* kv, int qk_bf16, const float* rpb, const float* out, const float* lse,
                   const float* d_out, float* d_q, float* d_k, float* d_rpb, void* ws, size_t ws_bytes, int B, int D,
                   int H, int W, int heads, int hd, float scale, modet_stream_t stream) {
  const float* q = (const float*)qv;
  const float* k = (const float*)kv;
  MODET_CHECK_PTR(q); MODET_CHECK_PTR(k); MODET_CHECK_PTR(rpb); MODET_CHECK_PTR(out); MODET_CHECK_PTR(lse);
  MODET_CHECK_PTR(d_out); MODET_CHECK_PTR(d_q); MODET_CHECK_PTR(d_k); MODET_CHECK_PTR(ws);        // (d_rpb may be NULL: header)
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && heads > 0);
  if (hd != HD && (qk_bf16 || !gen_hd_ok(hd))) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_na_bwd_ws_bytes(B, D, H, W, heads)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const NaMarchPlan mp = na_march_plan(B, D, H, W, heads, hd);
  if (mp.on) {
    const int64_t nblk = (int64_t)mp.tiles_x * mp.tiles_y * mp.nchunk;
    NaMarchArgs a{q, k, rpb, out, lse, d_out, d_q, d_k, (float*)ws, D, H, W, heads, mp.tiles_x, mp.tiles_y, mp.nchunk, mp.zc, scale};
    if (qk_bf16) hipLaunchKernelGGL(na_bwd_march_kernel<true>, dim3((unsigned)nblk, heads, B), dim3(NTHREADS), 0, s, a);
    else hipLaunchKernelGGL(na_bwd_march_kernel<false>, dim3((unsigned)nblk, heads, B), dim3(NTHREADS), 0, s, a);
    size_t fl = (size_t)B * heads * nblk * 27;
    fl += fl & 1;
    if (d_rpb) drpb_reduce((float*)ws, ws, fl * sizeof(float), d_rpb, B, heads, nblk, s);
    return modet_launch_status();
  }
  const TileGeom g = geom(D, H, W);
  const int64_t nblk = (int64_t)g.tiles_x * g.tiles_y * g.tiles_z;
  float* part = (float*)ws;
  dim3 grid((unsigned)nblk, heads, B);
  if (hd == HD) {
    if (qk_bf16) hipLaunchKernelGGL(na_bwd_kernel<true>, grid, dim3(NTHREADS), 0, s, q, k, rpb, out, lse, d_out, d_q, d_k, part, D, H, W,
                                    heads, scale, g);
    else hipLaunchKernelGGL(na_bwd_kernel<false>, grid, dim3(NTHREADS), 0, s, q, k, rpb, out, lse, d_out, d_q, d_k, part, D, H, W,
                            heads, scale, g);
  } else {
    hipLaunchKernelGGL(na_bwd_gen_kernel, grid, dim3(NTHREADS), 0, s, q, k, rpb, out, lse, d_out, d_q, d_k, part, D, H, W,
                       heads, hd, scale, g);
  }
  size_t fl = (size_t)B * heads * nblk * 27;
  fl += fl & 1;
  if (d_rpb) drpb_reduce(part, ws, fl * sizeof(float), d_rpb, B, heads, nblk, s);
  return modet_launch_status();
}

int64_t modet_na_bwd_partial_rows(int B, int D, int H, int W, int heads, int hd) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || heads <= 0) return 0;
  const NaMarchPlan mp = na_march_plan(B, D, H, W, heads, hd);
  if (mp.on) return (int64_t)mp.tiles_x * mp.tiles_y * mp.nchunk;
  const TileGeom g = geom(D, H, W);
  return (int64_t)g.tiles_x * g.tiles_y * g.tiles_z;
}

int modet_na_bwd(const float* q, const float* k, const float* rpb, const float* out, const float* lse,
                 const float* d_out, float* d_q, float* d_k, float* d_rpb, void* ws, size_t ws_bytes, int B, int D,
                 int H, int W, int heads, int hd, float scale, modet_stream_t stream) {
  return modet_na_bwd_t(q, k, 0, rpb, out, lse, d_out, d_q, d_k, d_rpb, ws, ws_bytes, B, D, H, W, heads, hd, scale, stream);
}

}  // extern "C"
