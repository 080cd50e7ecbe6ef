// The operator boundary: modet_fw / modet_bw with the CUDA op's exact tensor contract (ModeT-cu/modet/modet.cpp:4-31 ->
// modet_kernel.cu:17-381), float and double.
#include "common.h"
#include "drpb_reduce.h"

namespace {

// ------------------------------------------------------------------------------------------ reference contract
// one thread per (b,h,voxel); 256 consecutive voxels per workgroup so the (voxel,27) slab it produces is one
// contiguous 27 KB range of attn, written/read through LDS in fully coalesced rows.
constexpr int QK_BLOCK = 256;

template <typename T>
__device__ __forceinline__ void qk_decode(int64_t v, int H, int W, T& z, T& y, T& x) {
  x = (T)(v % W);
  const int64_t t = v / W;
  y = (T)(t % H);
  z = (T)(t / H);
}

// the operator dispatches float and double like the reference (AT_DISPATCH_FLOATING_TYPES, modet_kernel.cu:134,:364)
__device__ __forceinline__ float tfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double tfma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float twave_sum(float v) { return wave_sum(v); }
__device__ __forceinline__ double twave_sum(double v) { return wave_sum_d(v); }

template <typename T>
__global__ __launch_bounds__(QK_BLOCK) void qk_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kpad,
                                                          const T* __restrict__ rpb, T* __restrict__ attn,
                                                          int heads, int D, int H, int W, int hd) {
  __shared__ T slab[QK_BLOCK * 27];
  const int64_t V = (int64_t)D * H * W;
  const int bh = blockIdx.y, h = bh % heads;
  const int64_t v0 = (int64_t)blockIdx.x * QK_BLOCK, v = v0 + threadIdx.x;
  const int Hp = H + 2, Wp = W + 2;
  const int64_t Vp = (int64_t)(D + 2) * Hp * Wp;
  if (v < V) {
    int z, y, x;
    qk_decode(v, H, W, z, y, x);
    const T* qp = q + ((int64_t)bh * V + v) * hd;
    for (int ki = 0; ki < 3; ++ki)
      for (int kj = 0; kj < 3; ++kj)
        for (int kk = 0; kk < 3; ++kk) {
          const T* kp = kpad + ((int64_t)bh * Vp + ((int64_t)(z + ki) * Hp + (y + kj)) * Wp + (x + kk)) * hd;
          T s = (T)0;
          for (int c = 0; c < hd; ++c) s = tfma(qp[c], kp[c], s);
          const int t = ki * 9 + kj * 3 + kk;
          slab[threadIdx.x * 27 + t] = s + (rpb ? rpb[h * 27 + t] : (T)0);
        }
  }
  __syncthreads();
  const int64_t nvalid = (V - v0 < QK_BLOCK ? V - v0 : QK_BLOCK) * 27;
  T* dst = attn + ((int64_t)bh * V + v0) * 27;
  for (int64_t i = threadIdx.x; i < nvalid; i += QK_BLOCK) dst[i] = slab[i];
}

template <typename T>
__global__ __launch_bounds__(QK_BLOCK) void qk_dq_kernel(const T* __restrict__ dattn, const T* __restrict__ kpad,
                                                         T* __restrict__ dq, int D, int H, int W, int hd) {
  __shared__ T slab[QK_BLOCK * 27];
  const int64_t V = (int64_t)D * H * W;
  const int bh = blockIdx.y;
  const int64_t v0 = (int64_t)blockIdx.x * QK_BLOCK, v = v0 + threadIdx.x;
  const int64_t nvalid = (V - v0 < QK_BLOCK ? V - v0 : QK_BLOCK) * 27;
  const T* src = dattn + ((int64_t)bh * V + v0) * 27;
  for (int64_t i = threadIdx.x; i < nvalid; i += QK_BLOCK) slab[i] = src[i];
  __syncthreads();
  if (v >= V) return;
  const int Hp = H + 2, Wp = W + 2;
  const int64_t Vp = (int64_t)(D + 2) * Hp * Wp;
  int z, y, x;
  qk_decode(v, H, W, z, y, x);
  T* dqp = dq + ((int64_t)bh * V + v) * hd;
  for (int c = 0; c < hd; ++c) {
    T s = (T)0;
    for (int ki = 0; ki < 3; ++ki)
      for (int kj = 0; kj < 3; ++kj)
        for (int kk = 0; kk < 3; ++kk)
          s = tfma(slab[threadIdx.x * 27 + ki * 9 + kj * 3 + kk],
                   kpad[((int64_t)bh * Vp + ((int64_t)(z + ki) * Hp + (y + kj)) * Wp + (x + kk)) * hd + c], s);
    dqp[c] = s;
  }
}

// gather over the PADDED key volume (pad ring included, as modetdk_bw_kernel :209-267 returns it)
template <typename T>
__global__ __launch_bounds__(QK_BLOCK) void qk_dk_kernel(const T* __restrict__ dattn, const T* __restrict__ q,
                                                         T* __restrict__ dkpad, int D, int H, int W, int hd) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t Vp = (int64_t)(D + 2) * Hp * Wp, V = (int64_t)D * H * W;
  const int bh = blockIdx.y;
  const int64_t pv = (int64_t)blockIdx.x * QK_BLOCK + threadIdx.x;
  if (pv >= Vp) return;
  int pz, py, px;
  qk_decode(pv, Hp, Wp, pz, py, px);
  T* dkp = dkpad + ((int64_t)bh * Vp + pv) * hd;
  for (int c = 0; c < hd; ++c) {
    T s = (T)0;
    for (int ki = 0; ki < 3; ++ki) {
      const int z = pz - ki;
      if (z < 0 || z >= D) continue;
      for (int kj = 0; kj < 3; ++kj) {
        const int y = py - kj;
        if (y < 0 || y >= H) continue;
        for (int kk = 0; kk < 3; ++kk) {
          const int x = px - kk;
          if (x < 0 || x >= W) continue;
          const int64_t n = (int64_t)bh * V + ((int64_t)z * H + y) * W + x;
          s = tfma(q[n * hd + c], dattn[n * 27 + ki * 9 + kj * 3 + kk], s);
        }
      }
    }
    dkp[c] = s;
  }
}

// d_rpb partials: workgroup (chunk, b*heads+h) sums its 256*QK_RPB_ITERS voxels for all 27 tokens
constexpr int QK_RPB_ITERS = 16;
template <typename T>
__global__ __launch_bounds__(QK_BLOCK) void qk_drpb_partial_kernel(const T* __restrict__ dattn,
                                                                   T* __restrict__ part, int64_t V) {
  __shared__ T red[27 * (QK_BLOCK / 64)];
  const int bh = blockIdx.y;
  T acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (T)0;
  const int64_t base = (int64_t)blockIdx.x * QK_BLOCK * QK_RPB_ITERS;
  for (int it = 0; it < QK_RPB_ITERS; ++it) {
    const int64_t v = base + (int64_t)it * QK_BLOCK + threadIdx.x;
    if (v < V) {
      const T* p = dattn + ((int64_t)bh * V + v) * 27;
#pragma unroll
      for (int t = 0; t < 27; ++t) acc[t] += p[t];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const T r = twave_sum(acc[t]);
    if (lane == 0) red[wv * 27 + t] = r;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    T r = (T)0;
    for (int w = 0; w < QK_BLOCK / 64; ++w) r += red[w * 27 + threadIdx.x];
    part[((int64_t)bh * gridDim.x + blockIdx.x) * 27 + threadIdx.x] = r;
  }
}

// ------------------------------------------------------------------------------------------ plane-marching kernels
// The fast path of the operator (even head_dim 4/6/8; the model always calls it with 6, ModeT-cu/models.py:323).
// A workgroup owns an 8x32 (y,x) tile and marches along z over a chunk of planes; everything it needs more than once
// lives in LDS, everything in HBM is touched in contiguous row segments:
//   forward : three rolling kpad planes (tile + 1-voxel halo) in LDS; thread = voxel, 27 logits in registers, written
//             through a per-wave (64 voxel x 27) LDS slab so that attn leaves in contiguous 256-byte pieces.
//   backward: one march for d_q, d_kpad and the d_rpb partials (qk_bwd_plane_kernel below).
constexpr int OTY = 8, OTX = 32, ONT = OTY * OTX;
constexpr int OHY = OTY + 2, OHX = OTX + 2, OCELLS = OHY * OHX;   // 340 cells
constexpr int OROW27 = OHX * 27;                                   // d_attn elements of one staged row (918)

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

// Staging loads go through a buffer descriptor of ONE plane (wave-uniform base, 32-bit per-thread byte offset held in a
// register across the whole march): an offset at or past num_records returns 0, which is exactly the zero fill the halo
// cells outside the volume need -- no 64-bit address pair and no select per load.
using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned OP_OOB = 0x80000000u;             // planes are < 2 GiB (checked on the host)
template <typename T>
__device__ __forceinline__ BufRsrc plane_rsrc(const T* base, unsigned bytes) {
  // uniform by construction (kernel arguments and blockIdx only); readfirstlane makes that provable, otherwise every
  // buffer op is wrapped in a waterfall loop
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);   // the builtin returns a signed int
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float buf_ld1(BufRsrc r, unsigned off, const float*) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ double buf_ld1(BufRsrc r, unsigned off, const double*) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ float2 buf_ld2(BufRsrc r, unsigned off, const float*) {
  return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ double2 buf_ld2(BufRsrc r, unsigned off, const double*) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

struct OpPlan { int tiles_y, tiles_x, zlen, zchunks; };
// nz planes are split into chunks so that about `want_wgs` workgroups exist; every chunk re-stages two planes
inline OpPlan op_plan(int BH, int nz, int ny, int nx, int want_wgs, int min_len) {
  OpPlan p;
  p.tiles_y = cdiv(ny, OTY); p.tiles_x = cdiv(nx, OTX);
  const int64_t wg = (int64_t)BH * p.tiles_y * p.tiles_x;
  int c = (int)cdiv64(want_wgs, wg);
  const int maxc = nz / min_len > 0 ? nz / min_len : 1;
  c = c < 1 ? 1 : (c > maxc ? maxc : c);
  p.zlen = cdiv(nz, c);
  p.zchunks = cdiv(nz, p.zlen);
  return p;
}
constexpr int OP_WGS_FWD = 2048, OP_WGS_DK = 1024, OP_MIN_FWD = 8, OP_MIN_DK = 16;

// Staging of one plane tile of a (.., y, x, HD) tensor -- OHY rows of OHX cells -- through registers, one buffer
// descriptor per row: the descriptor's num_records is the valid length of that row segment (0 for a row outside the
// volume), so the hardware range check supplies every zero the halo needs and the thread keeps ONE offset register.
// `lo` cells at the start of every row lie outside the volume on the low side (d_kpad tiles only); they are zero-filled
// in LDS once and skipped here.
template <typename T, int HD>
struct RowStage {
  using V2 = typename Vec2<T>::type;
  static constexpr int ROWV = OHX * HD / 2;                 // 2-element vectors per staged row
  static constexpr int GROUPS = ROWV <= ONT / 2 ? 2 : 1, TPG = ONT / GROUPS, RPT = OHY / GROUPS;
  static_assert(ROWV <= TPG && OHY % GROUPS == 0, "row does not fit the thread group");
  V2 reg[RPT];
  int grp, idx;                                            // this thread's row group and vector index inside a row
  __device__ __forceinline__ void init() { grp = threadIdx.x / TPG; idx = threadIdx.x - grp * TPG; }
  // plane: first element of the plane; row r starts at element ((y_first + r) * pitch_cells + x_first) * HD; rows with
  // y outside [0, ny) are empty; `cells` valid cells per row from x_first on
  __device__ __forceinline__ void load(const T* __restrict__ plane, int y_first, int ny, int pitch_cells, int x_first, int cells) {
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int y = y_first + grp + m * GROUPS;
      const T* row = plane + ((int64_t)y * pitch_cells + x_first) * HD;
      const BufRsrc r = plane_rsrc(row, (y >= 0 && y < ny) ? (unsigned)(cells * HD * sizeof(T)) : 0u);
      reg[m] = buf_ld2(r, (unsigned)idx * (unsigned)sizeof(V2), (const T*)nullptr);
    }
  }
  __device__ __forceinline__ void store(T* __restrict__ slot, int lo) const {
    V2* s2 = reinterpret_cast<V2*>(slot);
    const int i = idx + lo * (HD / 2);
    if (i < ROWV) {
#pragma unroll
      for (int m = 0; m < RPT; ++m) s2[(grp + m * GROUPS) * ROWV + i] = reg[m];
    }
  }
};

template <typename T, int HD>
__device__ __forceinline__ void ldv(const T* __restrict__ p, T (&r)[HD]) {
  using V2 = typename Vec2<T>::type;
#pragma unroll
  for (int c = 0; c < HD / 2; ++c) {
    const V2 v = reinterpret_cast<const V2*>(p)[c];
    r[2 * c] = v.x; r[2 * c + 1] = v.y;
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(ONT, sizeof(T) == 4 ? 3 : 1) void qk_fwd_plane_kernel(const T* __restrict__ q, const T* __restrict__ kpad,
                                                           const T* __restrict__ rpb, T* __restrict__ attn, int heads,
                                                           int D, int H, int W, int tiles_x, int zlen) {
  using V2 = typename Vec2<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char op_smem[];
  T* kpl = reinterpret_cast<T*>(op_smem);            // [3][OCELLS*HD]
  T* slab = kpl + 3 * OCELLS * HD;                   // [ONT*27]
  const int tid = threadIdx.x, ty = tid / OTX, tx = tid - ty * OTX, lane = tid & 63, wv = tid >> 6;
  const int y0 = (blockIdx.x / tiles_x) * OTY, x0 = (blockIdx.x % tiles_x) * OTX;
  const int z0 = blockIdx.y * zlen, z1 = z0 + zlen < D ? z0 + zlen : D;
  const int bh = blockIdx.z, h = bh % heads;
  const int Hp = H + 2, Wp = W + 2;
  const int64_t V = (int64_t)D * H * W, planeV = (int64_t)Hp * Wp * (HD / 2);
  const V2* k2 = reinterpret_cast<const V2*>(kpad) + (int64_t)bh * (D + 2) * planeV;
  RowStage<T, HD> ks;
  ks.init();
  const int kcells = Wp - x0 < OHX ? Wp - x0 : OHX;
  auto kload = [&](int pz) { ks.load(reinterpret_cast<const T*>(k2 + (int64_t)pz * planeV), y0, Hp, Wp, x0, kcells); };
  kload(z0);
  ks.store(kpl + (z0 % 3) * OCELLS * HD, 0);
  kload(z0 + 1);
  ks.store(kpl + ((z0 + 1) % 3) * OCELLS * HD, 0);
  kload(z0 + 2);
  T rb[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) rb[t] = rpb ? rpb[h * 27 + t] : (T)0;
  const bool vox = (y0 + ty < H) && (x0 + tx < W);
  const int64_t vrow = ((int64_t)(y0 + ty) * W + x0 + tx);
  T qn[HD];
  const int64_t vq = vox ? vrow : 0;               // out-of-tile lanes read voxel 0 of the plane and never write
  ldv<T, HD>(q + ((int64_t)bh * V + (int64_t)z0 * H * W + vq) * HD, qn);
  const int nx27 = (W - x0 < OTX ? W - x0 : OTX) * 27;
  T* slw = slab + wv * 64 * 27;
  for (int z = z0; z < z1; ++z) {
    ks.store(kpl + ((z + 2) % 3) * OCELLS * HD, 0);
    __syncthreads();
    T qs[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) qs[c] = qn[c];
    if (z + 1 < z1) {
      kload(z + 3);
      ldv<T, HD>(q + ((int64_t)bh * V + (int64_t)(z + 1) * H * W + vq) * HD, qn);
    }
#pragma unroll
    for (int ki = 0; ki < 3; ++ki) {
      const T* pl = kpl + ((z + ki) % 3) * OCELLS * HD;
#pragma unroll
      for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          T kv[HD];
          ldv<T, HD>(pl + ((ty + kj) * OHX + tx + kk) * HD, kv);
          T s = (T)0;
#pragma unroll
          for (int c = 0; c < HD; ++c) s = tfma(qs[c], kv[c], s);
          const int t = ki * 9 + kj * 3 + kk;
          slw[lane * 27 + t] = s + rb[t];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the wave's two rows (2 x 32 voxels x 27) leave in contiguous pieces
    T* dst = attn + ((int64_t)bh * V + ((int64_t)z * H + y0 + 2 * wv) * W + x0) * 27;
#pragma unroll
    for (int j = 0; j < 27; ++j) {
      const int e = j * 64 + lane, r = e >= OTX * 27 ? 1 : 0, o = e - r * OTX * 27;
      if (y0 + 2 * wv + r < H && o < nx27) dst[(int64_t)r * W * 27 + o] = slw[e];
    }
    __syncthreads();
  }
}

// The whole backward in one march over SOURCE planes.  The tile is in PADDED (py,px) coordinates:
//   d_kpad : target (pz,py,px) gathers d_attn[z][y][x][t] * q[z][y][x] over z = pz-ki, y = py-kj, x = px-kk; plane z of
//            (d_attn, q) -- tile + 2-voxel halo on the low side -- is staged once and feeds target planes z, z+1, z+2,
//            whose accumulators rotate through registers; a target plane is written when its third source plane is done.
//   d_q    : for the plane just staged, the tile's own voxels (source (py0-1+ty, px0-1+tx)): d_attn straight from the
//            staged plane (stride 27 across lanes: conflict-free), three rolling kpad planes with origin (py0-1, px0-1).
//   d_rpb  : the thread's running sum of its own voxels' d_attn -> one partial row per workgroup (two-stage fp64 reduce).
// d_attn is read once (+ the (y,x) halo, 1.33x) for all three outputs, deterministically, no atomics.  (Measured against
// the same march split into a d_q and a d_kpad kernel: 0.364 vs 0.427 ms at 160x192x160, profiles/r02f_operator_*.json.)
template <typename T, int HD>
__global__ __launch_bounds__(ONT, sizeof(T) == 4 ? 2 : 1) void qk_bwd_plane_kernel(const T* __restrict__ dattn, const T* __restrict__ q,
                                                           const T* __restrict__ kpad, T* __restrict__ dq,
                                                           T* __restrict__ dkpad, T* __restrict__ part, int D, int H,
                                                           int W, int tiles_x, int zlen) {
  using V2 = typename Vec2<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char op_smem[];
  T* das = reinterpret_cast<T*>(op_smem);            // [OCELLS][27]
  T* qs = das + OCELLS * 27;                         // [OCELLS][HD]
  T* kpl = qs + OCELLS * HD;                         // [3][OCELLS][HD]
  constexpr int DRL = (OROW27 + ONT - 1) / ONT;
  const int tid = threadIdx.x, ty = tid / OTX, tx = tid - ty * OTX, lane = tid & 63, wv = tid >> 6;
  const int py0 = (blockIdx.x / tiles_x) * OTY, px0 = (blockIdx.x % tiles_x) * OTX;
  const int Dp = D + 2, Hp = H + 2, Wp = W + 2;
  const int pz0 = blockIdx.y * zlen, pz1 = pz0 + zlen < Dp ? pz0 + zlen : Dp;
  const int bh = blockIdx.z;
  const int64_t HW = (int64_t)H * W, V = (int64_t)D * HW, kplane = (int64_t)Hp * Wp * HD;
  const int lo = px0 == 0 ? 2 : 0, xs = px0 - 2 + lo;
  const int cells = (W - xs < OHX - lo ? W - xs : OHX - lo) > 0 ? (W - xs < OHX - lo ? W - xs : OHX - lo) : 0;
  const int klo = px0 == 0 ? 1 : 0, kxs = px0 - 1 + klo;
  const int kcells = Wp - kxs < OHX - klo ? Wp - kxs : OHX - klo;
  const T* dbase = dattn + (int64_t)bh * D * HW * 27;
  const T* qbase = q + (int64_t)bh * D * HW * HD;
  const T* kbase = kpad + (int64_t)bh * Dp * kplane;
  T dreg[OHY * DRL];
  RowStage<T, HD> qst, kst;
  qst.init(); kst.init();
  auto issue = [&](int z) {
#pragma unroll
    for (int ly = 0; ly < OHY; ++ly) {
      const int y = py0 - 2 + ly;
      const BufRsrc rs = plane_rsrc(dbase + (((int64_t)z * H + y) * W + xs) * 27,
                                    (y >= 0 && y < H) ? (unsigned)(cells * 27 * sizeof(T)) : 0u);
#pragma unroll
      for (int k = 0; k < DRL; ++k) dreg[ly * DRL + k] = buf_ld1(rs, (unsigned)(tid + ONT * k) * (unsigned)sizeof(T), (const T*)nullptr);
    }
    qst.load(qbase + (int64_t)z * HW * HD, py0 - 2, H, W, xs, cells);
  };
  auto kload = [&](int pz) { kst.load(kbase + (int64_t)pz * kplane, py0 - 1, Hp, Wp, kxs, kcells); };
  for (int e = tid; e < OCELLS * (27 + 4 * HD); e += ONT) das[e] = (T)0;       // das, qs and the three k slots
  __syncthreads();
  const int zs = pz0 - 2 > 0 ? pz0 - 2 : 0, dq_end = pz1 - 2;                   // d_q planes of this chunk: [zs, dq_end)
  if (zs < dq_end) {
    kload(zs);     kst.store(kpl + (zs % 3) * OCELLS * HD, klo);
    kload(zs + 1); kst.store(kpl + ((zs + 1) % 3) * OCELLS * HD, klo);
    kload(zs + 2);
  }
  T acc[3][HD], racc[27];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[i][c] = (T)0;
#pragma unroll
  for (int t = 0; t < 27; ++t) racc[t] = (T)0;
  const bool tgt = (py0 + ty < Hp) && (px0 + tx < Wp);
  const int vy = py0 - 1 + ty, vx = px0 - 1 + tx;
  const bool vox = vy >= 0 && vy < H && vx >= 0 && vx < W;
  const int own = (ty + 1) * OHX + tx + 1;
  int z = pz0 - 2;
  if (z >= 0 && z < D) issue(z);
  for (; z < pz1; ++z) {
    const bool src = z >= 0 && z < D, dqa = src && z < dq_end;        // uniform over the workgroup
    if (src) {
      __syncthreads();
#pragma unroll
      for (int ly = 0; ly < OHY; ++ly)
#pragma unroll
        for (int k = 0; k < DRL; ++k) {
          const int e = lo * 27 + tid + ONT * k;
          if (e < OROW27) das[ly * OROW27 + e] = dreg[ly * DRL + k];
        }
      qst.store(qs, lo);
      if (dqa) kst.store(kpl + ((z + 2) % 3) * OCELLS * HD, klo);
      __syncthreads();
    }
    if (z + 1 < pz1 && z + 1 >= 0 && z + 1 < D) issue(z + 1);
    if (dqa && z + 1 < dq_end) kload(z + 3);
    if (src) {
#pragma unroll 1
      for (int kj = 0; kj < 3; ++kj)               // a real loop: bounds how many LDS reads the scheduler hoists
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const int cell = (ty + 2 - kj) * OHX + tx + 2 - kk;
          T qv[HD];
          ldv<T, HD>(qs + cell * HD, qv);
#pragma unroll
          for (int ki = 0; ki < 3; ++ki) {
            const T a = das[cell * 27 + ki * 9 + kj * 3 + kk];
#pragma unroll
            for (int c = 0; c < HD; ++c) acc[ki][c] = tfma(qv[c], a, acc[ki][c]);
          }
        }
    }
    if (dqa) {
      T g[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) g[c] = (T)0;
#pragma unroll
      for (int t = 0; t < 27; ++t) racc[t] += das[own * 27 + t];      // cells outside the volume hold zeros
#pragma unroll 1
      for (int ki = 0; ki < 3; ++ki) {
        const T* pl = kpl + ((z + ki) % 3) * OCELLS * HD;
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const int t = ki * 9 + kj * 3 + kk;
            const T a = das[own * 27 + t];
            T kv[HD];
            ldv<T, HD>(pl + ((ty + kj) * OHX + tx + kk) * HD, kv);
#pragma unroll
            for (int c = 0; c < HD; ++c) g[c] = tfma(a, kv[c], g[c]);
          }
      }
      if (vox) {
        V2* o2 = reinterpret_cast<V2*>(dq + ((int64_t)bh * V + ((int64_t)z * H + vy) * W + vx) * HD);
#pragma unroll
        for (int c = 0; c < HD / 2; ++c) { V2 v; v.x = g[2 * c]; v.y = g[2 * c + 1]; o2[c] = v; }
      }
    }
    if (z >= pz0 && tgt) {
      V2* o2 = reinterpret_cast<V2*>(dkpad + ((((int64_t)bh * Dp + z) * Hp + py0 + ty) * Wp + px0 + tx) * HD);
#pragma unroll
      for (int c = 0; c < HD / 2; ++c) { V2 v; v.x = acc[0][2 * c]; v.y = acc[0][2 * c + 1]; o2[c] = v; }
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) { acc[0][c] = acc[1][c]; acc[1][c] = acc[2][c]; acc[2][c] = (T)0; }
  }
  if (part) {
    __syncthreads();
    T* red = das;                        // [4][27]
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const T r = twave_sum(racc[t]);
      if (lane == 0) red[wv * 27 + t] = r;
    }
    __syncthreads();
    if (tid < 27) {
      const int64_t nblk = (int64_t)gridDim.x * gridDim.y, blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
      part[((int64_t)bh * nblk + blk) * 27 + tid] = ((red[tid] + red[27 + tid]) + red[54 + tid]) + red[81 + tid];
    }
  }
}

template <typename T, int HD> constexpr size_t op_lds_fwd() { return (size_t)(3 * OCELLS * HD + ONT * 27) * sizeof(T); }
template <typename T, int HD> constexpr size_t op_lds_bwd() { return (size_t)(OCELLS * (27 + 4 * HD)) * sizeof(T); }
inline bool op_plane_ok(int hd, int H, int W, size_t elem) {     // even small head_dim, every staged plane below 2 GiB
  return (hd == 4 || hd == 6 || hd == 8) && (uint64_t)(H + 2) * (W + 2) * 27 * elem < 0x80000000ull;
}

template <typename K>
inline void op_allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T, int HD>
inline void op_fwd_launch(const T* q, const T* kpad, const T* rpb, T* attn, int B, int heads, int D, int H, int W, hipStream_t s) {
  const OpPlan p = op_plan(B * heads, D, H, W, OP_WGS_FWD, OP_MIN_FWD);
  constexpr size_t lds = op_lds_fwd<T, HD>();
  op_allow_lds(qk_fwd_plane_kernel<T, HD>, lds);
  hipLaunchKernelGGL((qk_fwd_plane_kernel<T, HD>), dim3(p.tiles_y * p.tiles_x, p.zchunks, B * heads), dim3(ONT), lds, s, q, kpad,
                     rpb, attn, heads, D, H, W, p.tiles_x, p.zlen);
}
inline int64_t op_dq_nblk(int BH, int D, int H, int W) {      // d_rpb partial rows per (b, head)
  const OpPlan p = op_plan(BH, D + 2, H + 2, W + 2, OP_WGS_DK, OP_MIN_DK);
  return (int64_t)p.tiles_y * p.tiles_x * p.zchunks;
}
template <typename T, int HD>
inline void op_bwd_launch(const T* d_attn, const T* q, const T* kpad, T* d_q, T* d_kpad, T* part, int B, int heads, int D, int H,
                          int W, hipStream_t s) {
  const OpPlan k = op_plan(B * heads, D + 2, H + 2, W + 2, OP_WGS_DK, OP_MIN_DK);
  constexpr size_t lb = op_lds_bwd<T, HD>();
  op_allow_lds(qk_bwd_plane_kernel<T, HD>, lb);
  hipLaunchKernelGGL((qk_bwd_plane_kernel<T, HD>), dim3(k.tiles_y * k.tiles_x, k.zchunks, B * heads), dim3(ONT), lb, s, d_attn,
                     q, kpad, d_q, d_kpad, part, D, H, W, k.tiles_x, k.zlen);
}

// float / double bodies of the operator boundary (the reference dispatches both, modet_kernel.cu:134,:364)
template <typename T>
int qk_fwd_impl(const T* q, const T* kpad, const T* rpb, T* attn, int B, int heads, int D, int H, int W, int hd,
                modet_stream_t stream) {
  MODET_CHECK_PTR(q); MODET_CHECK_PTR(kpad); MODET_CHECK_PTR(attn);
  MODET_CHECK_DIM(B > 0 && heads > 0 && hd > 0);
  MODET_CHECK_DIM(D >= 3 && H >= 3 && W >= 3);   // CHECK_3DFEATMAP, utils.h:10
  if (B * heads > 65535) return MODET_ERR_DIM;
  hipStream_t s = (hipStream_t)stream;
  switch (op_plane_ok(hd, H, W, sizeof(T)) ? hd : 0) {
    case 4: op_fwd_launch<T, 4>(q, kpad, rpb, attn, B, heads, D, H, W, s); return modet_launch_status();
    case 6: op_fwd_launch<T, 6>(q, kpad, rpb, attn, B, heads, D, H, W, s); return modet_launch_status();
    case 8: op_fwd_launch<T, 8>(q, kpad, rpb, attn, B, heads, D, H, W, s); return modet_launch_status();
    default: break;                                     // any other head_dim: the one-thread-per-voxel kernels
  }
  const int64_t V = (int64_t)D * H * W;
  dim3 grid((unsigned)cdiv64(V, QK_BLOCK), B * heads);
  hipLaunchKernelGGL(qk_fwd_kernel<T>, grid, dim3(QK_BLOCK), 0, (hipStream_t)stream, q, kpad, rpb, attn, heads, D, H, W,
                     hd);
  return modet_launch_status();
}

inline size_t qk_ws_bytes(int B, int heads, int D, int H, int W, size_t elem) {
  const int64_t V = (int64_t)D * H * W;
  int64_t rows = cdiv64(V, (int64_t)QK_BLOCK * QK_RPB_ITERS);         // generic path
  const int64_t prow = op_dq_nblk(B * heads, D, H, W);                // plane-marching path (hd 4/6/8)
  if (prow > rows) rows = prow;
  size_t fl = (size_t)B * heads * rows * 27;
  fl += fl & 1;                                        // keep the fp64 scratch that follows 8-byte aligned
  return fl * elem + drpb_scratch_bytes(B, heads);
}

template <typename T>
int qk_bwd_impl(const T* d_attn, const T* q, const T* kpad, T* d_q, T* d_kpad, T* d_rpb, void* ws, size_t ws_bytes,
                int B, int heads, int D, int H, int W, int hd, modet_stream_t stream) {
  MODET_CHECK_PTR(d_attn); MODET_CHECK_PTR(q); MODET_CHECK_PTR(kpad); MODET_CHECK_PTR(d_q); MODET_CHECK_PTR(d_kpad);
  MODET_CHECK_DIM(B > 0 && heads > 0 && hd > 0);
  MODET_CHECK_DIM(D >= 3 && H >= 3 && W >= 3);
  hipStream_t s = (hipStream_t)stream;
  const int64_t V = (int64_t)D * H * W, Vp = (int64_t)(D + 2) * (H + 2) * (W + 2);
  if (B * heads > 65535) return MODET_ERR_DIM;
  if (d_rpb) {
    MODET_CHECK_PTR(ws);
    if (ws_bytes < qk_ws_bytes(B, heads, D, H, W, sizeof(T))) return MODET_ERR_WORKSPACE;
  }
  if (op_plane_ok(hd, H, W, sizeof(T))) {
    T* part = d_rpb ? (T*)ws : nullptr;
    switch (hd) {
      case 4: op_bwd_launch<T, 4>(d_attn, q, kpad, d_q, d_kpad, part, B, heads, D, H, W, s); break;
      case 6: op_bwd_launch<T, 6>(d_attn, q, kpad, d_q, d_kpad, part, B, heads, D, H, W, s); break;
      default: op_bwd_launch<T, 8>(d_attn, q, kpad, d_q, d_kpad, part, B, heads, D, H, W, s); break;
    }
    if (d_rpb) {
      const int64_t nblk = op_dq_nblk(B * heads, D, H, W);
      size_t fl = (size_t)B * heads * nblk * 27;
      fl += fl & 1;
      drpb_reduce<T>((const T*)ws, ws, fl * sizeof(T), d_rpb, B, heads, nblk, s);
    }
    return modet_launch_status();
  }
  if (d_rpb) {
    const int64_t nchunk = cdiv64(V, (int64_t)QK_BLOCK * QK_RPB_ITERS);
    hipLaunchKernelGGL(qk_drpb_partial_kernel<T>, dim3((unsigned)nchunk, B * heads), dim3(QK_BLOCK), 0, s, d_attn,
                       (T*)ws, V);
    size_t fl = (size_t)B * heads * nchunk * 27;
    fl += fl & 1;
    drpb_reduce<T>((const T*)ws, ws, fl * sizeof(T), d_rpb, B, heads, nchunk, s);
  }
  hipLaunchKernelGGL(qk_dq_kernel<T>, dim3((unsigned)cdiv64(V, QK_BLOCK), B * heads), dim3(QK_BLOCK), 0, s, d_attn,
                     kpad, d_q, D, H, W, hd);
  hipLaunchKernelGGL(qk_dk_kernel<T>, dim3((unsigned)cdiv64(Vp, QK_BLOCK), B * heads), dim3(QK_BLOCK), 0, s, d_attn, q,
                     d_kpad, D, H, W, hd);
  return modet_launch_status();
}

}  // namespace

extern "C" {

int modet_qk_fwd(const float* q, const float* kpad, const float* rpb, float* attn, int B, int heads, int D, int H,
                 int W, int hd, modet_stream_t stream) {
  return qk_fwd_impl<float>(q, kpad, rpb, attn, B, heads, D, H, W, hd, stream);
}
int modet_qk_fwd_f64(const double* q, const double* kpad, const double* rpb, double* attn, int B, int heads, int D,
                     int H, int W, int hd, modet_stream_t stream) {
  return qk_fwd_impl<double>(q, kpad, rpb, attn, B, heads, D, H, W, hd, stream);
}

size_t modet_qk_bwd_ws_bytes(int B, int heads, int D, int H, int W) { return qk_ws_bytes(B, heads, D, H, W, sizeof(float)); }
size_t modet_qk_bwd_ws_bytes_f64(int B, int heads, int D, int H, int W) {
  return qk_ws_bytes(B, heads, D, H, W, sizeof(double));
}

int modet_qk_bwd(const float* d_attn, const float* q, const float* kpad, float* d_q, float* d_kpad, float* d_rpb,
                 void* ws, size_t ws_bytes, int B, int heads, int D, int H, int W, int hd, modet_stream_t stream) {
  return qk_bwd_impl<float>(d_attn, q, kpad, d_q, d_kpad, d_rpb, ws, ws_bytes, B, heads, D, H, W, hd, stream);
}
int modet_qk_bwd_f64(const double* d_attn, const double* q, const double* kpad, double* d_q, double* d_kpad,
                     double* d_rpb, void* ws, size_t ws_bytes, int B, int heads, int D, int H, int W, int hd,
                     modet_stream_t stream) {
  return qk_bwd_impl<double>(d_attn, q, kpad, d_q, d_kpad, d_rpb, ws, ws_bytes, B, heads, D, H, W, hd, stream);
}

}  // extern "C"
