// The trilinear warp's backward WITHOUT global float atomics (SpatialTransformer backward, reference ModeT/models.py:55-67 ->
// ATen grid_sampler_3d_backward scatters d_src with atomicAdd).  Default path of every warp of the model whose backward scatters
// (feature warps: C % 8 == 0, fp32 or bf16 src; the unbounded flow compositions: C == 3 with add_flow) since round 6; design,
// measurements and the rejected variants: DESIGN.md section 4.2, profiles/r06b_warp_tile_experiments.txt.
//   Why tiles: the float-atomic scatter is bound by the L2 atomic unit (18 G sector-atomics/s on the model's flow, 3.3 x write
//   amplification) and is not reproducible; ds_add_f32 retires 0.33 lanes/clk/CU, but the INTEGER LDS atomics run 20-37 x faster
//   (profiles/r05y_lds_atomic_microbench.txt).
//   FILL        source voxels are binned by the 8^3 DESTINATION tile of their base corner: per source workgroup (SZ x 8 x 32 voxels)
//               an LDS hash histogram, one returning global atomic per (workgroup, tile) reserves positions in the tile's list.
//               Every tile owns a FIXED segment of CAP entries (three times what a non-folding flow sends it); entries beyond it
//               go to one overflow list.  A voxel whose d_out is all zero or whose corners all leave the volume is dropped; the
//               others get a PAYLOAD entry -- (voxel, flow, d_out) -- so the next pass chases no index, and their d_flow (the eight
//               src corners gathered through L2: this pass is a stream with >= 20 waves per CU, where a gather's latency hides).
//               max |d_out| of the kept entries falls out of the same read (one atomicMax per workgroup over 64 slots).
//   ACCUMULATE  one 512-lane workgroup per (tile, group of 8 channels): the tile's payload, a contiguous stream, goes into a
//               9^3 x 8 window of 64-bit FIXED-POINT sums in LDS (2^-40 of the power of two above max |d_out| per unit, scaled
//               per entry), laid out [channel][cell].  46 KB: three workgroups per CU.  The 8^3 owned cells leave as plain stores
//               (d_src needs no zero fill), the 217 high-face cells go to a side buffer.  A tile without entries stores zeros.
//               Entries of the overflow list (usually none) are filtered by tile id and added the same way.
//   BORDER      every owned cell on a low tile face adds the high-face cells of the (non-empty) neighbours, in a fixed order.
// Integer sums: the result does not depend on the order of the lists -> bit-reproducible run to run.  Every launch is a kernel
// with fixed arguments (no memset node, no host read): capturable into a hipGraph.  Non-finite d_out: max |d_out| is taken on
// bit patterns (NaN orders above inf), and a non-finite maximum poisons d_src with NaN instead of converting garbage.
#include "common.h"

namespace {
constexpr int TL = 8, WN = 9, CELLS = WN * WN * WN, NBORDER = CELLS - TL * TL * TL;      // 729 window cells, 217 on the high faces
constexpr int SY = 8, SX = 32;                                // source block of the fill pass: SZ x 8 x 32 voxels, SZ per thread; SZ = 4, or
                                                              // 1 below 400 k voxels (levels 3-5: 4 x more workgroups -- latency-, not bandwidth-bound there)
// a source block holds SZ x 256 voxels, so at most that many distinct tiles: with twice the slots (HASH = 512 SZ) linear probing
// always ends on a free slot or on the key (round 5's 256 slots for 1024 voxels overflowed silently on flows rougher than ~12
// voxels: ADVICE r5)
constexpr int ACC = 512;                                      // lanes of an accumulate workgroup
constexpr int FXBITS = 40;
struct Geo { int D, H, W, C, tz, ty, tx, ntiles, B, dbg; };      // ntiles = tiles per sample; dbg: tuning builds only

struct Entry { int tile, bz, by, bx; float fz, fy, fx; };
// the ONE place that decides a voxel's base cell and tile: the fill and the accumulate pass must agree bit for bit
__device__ __forceinline__ bool make_entry(float f0, float f1, float f2, int z, int y, int x, const Geo& g, Entry& e) {
  const float pz = (float)z + f0, py = (float)y + f1, px = (float)x + f2;
  const float flz = floorf(pz), fly = floorf(py), flx = floorf(px);
  if (!(flz >= -1.f && flz <= (float)(g.D - 1) && fly >= -1.f && fly <= (float)(g.H - 1) && flx >= -1.f && flx <= (float)(g.W - 1)))
    return false;                                   // every corner outside the volume (or a non-finite flow)
  e.bz = (int)flz; e.by = (int)fly; e.bx = (int)flx;
  e.fz = pz - flz; e.fy = py - fly; e.fx = px - flx;
  const int cz = (e.bz < 0 ? 0 : e.bz) >> 3, cy = (e.by < 0 ? 0 : e.by) >> 3, cx = (e.bx < 0 ? 0 : e.bx) >> 3;
  e.tile = (cz * g.ty + cy) * g.tx + cx;
  return true;
}

template <int HASH>
__device__ __forceinline__ int hash_insert(int* keys, unsigned* cnt, int tile, unsigned& rank) {
  int s = (int)(((unsigned)tile * 40503u) & (unsigned)(HASH - 1));
  for (;;) {                                        // (terminates: at most HASH / 2 keys)
    const int k0 = atomicCAS(&keys[s], -1, tile);
    if (k0 == -1 || k0 == tile) break;
    s = (s + 1) & (HASH - 1);
  }
  rank = atomicAdd(&cnt[s], 1u);
  return s;
}

template <int SZ>
__device__ __forceinline__ void block_origin(int blk, int bx_n, int by_n, int& x0, int& y0, int& z0) {
  x0 = (blk % bx_n) * SX; blk /= bx_n;
  y0 = (blk % by_n) * SY;
  z0 = (blk / by_n) * SZ;
}

constexpr unsigned CAP = 1536;                                // entries of a tile's own list segment
// where the entries go: tile t owns list[t * CAP ... ) (cursor[t] = entries handed out so far, may exceed CAP), the rest goes to
// ovf_list[ovf_count++] with its tile id beside it
struct Lists { float* list; unsigned* cursor; unsigned* ovf_count; unsigned* ovf_tile; float* ovf_list; };
__device__ __forceinline__ float* entry_slot(const Lists& L, int tile, unsigned pos, int S) {
  if (pos < CAP) return L.list + ((size_t)tile * CAP + pos) * S;
  const unsigned oi = atomicAdd(L.ovf_count, 1u);
  L.ovf_tile[oi] = (unsigned)tile;
  return L.ovf_list + (size_t)oi * S;
}

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// eight channels of src (fp32, or bf16 widened) at element offset `e`
template <bool S16>
__device__ __forceinline__ void load8(const void* __restrict__ src, int64_t e, float (&s)[8]) {
  if constexpr (S16) {
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(src) + e);
    s[0] = bf16_lo(u.x); s[1] = bf16_hi(u.x); s[2] = bf16_lo(u.y); s[3] = bf16_hi(u.y);
    s[4] = bf16_lo(u.z); s[5] = bf16_hi(u.z); s[6] = bf16_lo(u.w); s[7] = bf16_hi(u.w);
  } else {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + e);
    const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + e + 4);
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = c.x; s[5] = c.y; s[6] = c.z; s[7] = c.w;
  }
}

// ---- B: cursor[tile] (zero on entry) hands out positions per (workgroup, tile); kept voxels write their
// payload entry [voxel id, flow x 3, d_out x C] there; d_flow of EVERY voxel of the block is written here (dropped: d_flow_add or 0)
template <bool S16, int SZ>
__global__ __launch_bounds__(256) void fill_kernel(const void* __restrict__ src, const float* __restrict__ flow,
                                                   const float* __restrict__ dout, const Lists L,
                                                   unsigned* __restrict__ amax, float* __restrict__ dflow,
                                                   const float* __restrict__ dflow_add, const Geo g, int bx_n, int by_n, int dbg_arg) {
#ifdef MODET_TUNING
  const int dbg = dbg_arg;
#else
  constexpr int dbg = 0; (void)dbg_arg;
#endif
  constexpr int HASH = 512 * SZ;
  __shared__ int keys[HASH];
  __shared__ unsigned cnt[HASH], off[HASH];
  __shared__ unsigned wmax[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < HASH; i += 256) { keys[i] = -1; cnt[i] = 0; }
  __syncthreads();
  int x0, y0, z0;
  block_origin<SZ>(blockIdx.x, bx_n, by_n, x0, y0, z0);
  const int x = x0 + (tid & 31), y = y0 + (tid >> 5);
  const int b = blockIdx.y;
  const int64_t V = (int64_t)g.D * g.H * g.W;
  const int C = g.C;
  flow += b * V * 3;
  dout += b * V * C;
  if (dflow) dflow += b * V * 3;
  if (dflow_add) dflow_add += b * V * 3;
  const int64_t sbase = b * V * C;                            // element offset of the sample inside src
  const int64_t sX = C, sY = (int64_t)g.W * C, sZ = (int64_t)g.H * g.W * C;
  int slot[SZ];
  unsigned rank[SZ];
  float f[SZ][3];
  unsigned mbits = 0;
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    const int z = z0 + k;
    const bool in = x < g.W && y < g.H && z < g.D;
    const int64_t p = in ? ((int64_t)z * g.H + y) * g.W + x : 0;
    f[k][0] = flow[p * 3]; f[k][1] = flow[p * 3 + 1]; f[k][2] = flow[p * 3 + 2];
    slot[k] = in ? -1 : -2;
  }
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    if (slot[k] == -2) continue;
    const int z = z0 + k;
    const int64_t p = ((int64_t)z * g.H + y) * g.W + x;
    Entry e;
    const bool has = make_entry(f[k][0], f[k][1], f[k][2], z, y, x, g, e);
    float gz = 0.f, gy = 0.f, gx = 0.f;
    unsigned vb = 0;                                          // max of the bit patterns of |d_out[p, :]| (0 <=> all zero)
    const float* dp = dout + p * C;
    if (has) {
      for (int c0 = 0; c0 < C; c0 += 8) {
        const float4 ga = *reinterpret_cast<const float4*>(dp + c0), gb = *reinterpret_cast<const float4*>(dp + c0 + 4);
        const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) { const unsigned u = __float_as_uint(fabsf(gv[c])); vb = u > vb ? u : vb; }
      }
    }
    if (has && vb && dflow && !(dbg & 2)) {                   // the eight src corners, gathered (only where something arrives)
      // corner validity (the base cell may be -1, the upper corner may be the dimension)
      const bool okz[2] = {e.bz >= 0, e.bz + 1 < g.D}, oky[2] = {e.by >= 0, e.by + 1 < g.H}, okx[2] = {e.bx >= 0, e.bx + 1 < g.W};
      const int64_t o000 = sbase + (((int64_t)e.bz * g.H + e.by) * g.W + e.bx) * C;
      float dot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c0 = 0; c0 < C; c0 += 8) {
        const float4 ga = *reinterpret_cast<const float4*>(dp + c0), gb = *reinterpret_cast<const float4*>(dp + c0 + 4);
        const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool ok = okz[q >> 2] && oky[(q >> 1) & 1] && okx[q & 1];
          const int64_t o = o000 + ((q >> 2) ? sZ : 0) + (((q >> 1) & 1) ? sY : 0) + ((q & 1) ? sX : 0) + c0;
          float s[8];
          load8<S16>(src, ok ? o : sbase, s);                 // (unconditional load of a valid address; masked below)
          float d = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) d = fmaf(s[c], gv[c], d);
          dot[q] += ok ? d : 0.f;
        }
      }
      const float wz[2] = {1.f - e.fz, e.fz}, wy[2] = {1.f - e.fy, e.fy}, wx[2] = {1.f - e.fx, e.fx};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        gz += (dz ? 1.f : -1.f) * wy[dy] * wx[dx] * dot[q];
        gy += (dy ? 1.f : -1.f) * wz[dz] * wx[dx] * dot[q];
        gx += (dx ? 1.f : -1.f) * wz[dz] * wy[dy] * dot[q];
      }
    }
    if (dflow && !(dbg & 4)) {
      if (dflow_add) { gz += dflow_add[p * 3]; gy += dflow_add[p * 3 + 1]; gx += dflow_add[p * 3 + 2]; }
      dflow[p * 3] = gz; dflow[p * 3 + 1] = gy; dflow[p * 3 + 2] = gx;
    }
    if (!has || vb == 0) continue;                            // dropped: contributes nothing to d_src
    mbits = vb > mbits ? vb : mbits;
    slot[k] = hash_insert<HASH>(keys, cnt, e.tile + b * g.ntiles, rank[k]);
  }
  __syncthreads();
  for (int i = tid; i < HASH; i += 256)
    if (keys[i] >= 0) off[i] = atomicAdd(&L.cursor[keys[i]], cnt[i]);
  for (int o = 32; o; o >>= 1) { const unsigned v = __shfl_xor(mbits, o, 64); mbits = v > mbits ? v : mbits; }
  if ((tid & 63) == 0) wmax[tid >> 6] = mbits;
  __syncthreads();
  if (tid == 0) {                                             // one atomic per workgroup, spread over 64 addresses
    unsigned m = wmax[0];
    for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
    if (m) atomicMax(&amax[(blockIdx.x + blockIdx.y) & 63], m);
  }
  const int S = 4 + C;                                        // words per payload entry (a multiple of 4)
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    if (slot[k] < 0 || (dbg & 1)) continue;
    const int z = z0 + k;
    const int64_t p = ((int64_t)z * g.H + y) * g.W + x;
    float* dst = entry_slot(L, keys[slot[k]], off[slot[k]] + rank[k], S);
    *reinterpret_cast<float4*>(dst) = make_float4(__int_as_float((z << 20) | (y << 10) | x), f[k][0], f[k][1], f[k][2]);   // (dimensions <= 1024)
    const float* dp = dout + p * C;                           // (read a moment ago: L1 / L2)
    for (int c0 = 0; c0 < C; c0 += 4) *reinterpret_cast<float4*>(dst + 4 + c0) = *reinterpret_cast<const float4*>(dp + c0);
  }
}

// ---- B for C == 3 (the flow compositions whose flow is NOT bounded by one voxel: warp(up(2 flow), w) + w with w a CWM output,
// reference models.py:392-403; the bounded ones gather, warp.hip): three scalars per voxel, payload entry = [header][d_out x 3, 0]
// (S = 8 words); add_flow: out = warp(src, flow) + flow, so d_flow += d_out for EVERY voxel.
template <int SZ>
__global__ __launch_bounds__(256) void fill_c3_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                      const float* __restrict__ dout, const Lists L,
                                                      unsigned* __restrict__ amax, float* __restrict__ dflow,
                                                      const float* __restrict__ dflow_add, const Geo g, int bx_n, int by_n, int add_flow) {
  constexpr int HASH = 512 * SZ;
  __shared__ int keys[HASH];
  __shared__ unsigned cnt[HASH], off[HASH];
  __shared__ unsigned wmax[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < HASH; i += 256) { keys[i] = -1; cnt[i] = 0; }
  __syncthreads();
  int x0, y0, z0;
  block_origin<SZ>(blockIdx.x, bx_n, by_n, x0, y0, z0);
  const int x = x0 + (tid & 31), y = y0 + (tid >> 5);
  const int b = blockIdx.y;
  const int64_t V = (int64_t)g.D * g.H * g.W;
  flow += b * V * 3;
  dout += b * V * 3;
  src += b * V * 3;
  if (dflow) dflow += b * V * 3;
  if (dflow_add) dflow_add += b * V * 3;
  const int64_t sX = 3, sY = (int64_t)g.W * 3, sZ = (int64_t)g.H * g.W * 3;
  int slot[SZ];
  unsigned rank[SZ];
  float f[SZ][3], gv[SZ][3];
  unsigned mbits = 0;
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    const int z = z0 + k;
    const bool in = x < g.W && y < g.H && z < g.D;
    const int64_t p = in ? ((int64_t)z * g.H + y) * g.W + x : 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { f[k][a] = flow[p * 3 + a]; gv[k][a] = dout[p * 3 + a]; }
    slot[k] = in ? -1 : -2;
  }
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    if (slot[k] == -2) continue;
    const int z = z0 + k;
    const int64_t p = ((int64_t)z * g.H + y) * g.W + x;
    Entry e;
    const bool has = make_entry(f[k][0], f[k][1], f[k][2], z, y, x, g, e);
    float gz = 0.f, gy = 0.f, gx = 0.f;
    unsigned vb = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { const unsigned u = __float_as_uint(fabsf(gv[k][a])); vb = u > vb ? u : vb; }
    if (has && vb && dflow) {
      const bool okz[2] = {e.bz >= 0, e.bz + 1 < g.D}, oky[2] = {e.by >= 0, e.by + 1 < g.H}, okx[2] = {e.bx >= 0, e.bx + 1 < g.W};
      const int64_t o000 = (((int64_t)e.bz * g.H + e.by) * g.W + e.bx) * 3;
      const float wz[2] = {1.f - e.fz, e.fz}, wy[2] = {1.f - e.fy, e.fy}, wx[2] = {1.f - e.fx, e.fx};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        const bool ok = okz[dz] && oky[dy] && okx[dx];
        const int64_t o = ok ? o000 + (dz ? sZ : 0) + (dy ? sY : 0) + (dx ? sX : 0) : 0;
        float d = fmaf(src[o + 2], gv[k][2], fmaf(src[o + 1], gv[k][1], src[o] * gv[k][0]));
        d = ok ? d : 0.f;
        gz += (dz ? 1.f : -1.f) * wy[dy] * wx[dx] * d;
        gy += (dy ? 1.f : -1.f) * wz[dz] * wx[dx] * d;
        gx += (dx ? 1.f : -1.f) * wz[dz] * wy[dy] * d;
      }
    }
    if (dflow) {
      if (add_flow) { gz += gv[k][0]; gy += gv[k][1]; gx += gv[k][2]; }
      if (dflow_add) { gz += dflow_add[p * 3]; gy += dflow_add[p * 3 + 1]; gx += dflow_add[p * 3 + 2]; }
      dflow[p * 3] = gz; dflow[p * 3 + 1] = gy; dflow[p * 3 + 2] = gx;
    }
    if (!has || vb == 0) continue;
    mbits = vb > mbits ? vb : mbits;
    slot[k] = hash_insert<HASH>(keys, cnt, e.tile + b * g.ntiles, rank[k]);
  }
  __syncthreads();
  for (int i = tid; i < HASH; i += 256)
    if (keys[i] >= 0) off[i] = atomicAdd(&L.cursor[keys[i]], cnt[i]);
  for (int o = 32; o; o >>= 1) { const unsigned v = __shfl_xor(mbits, o, 64); mbits = v > mbits ? v : mbits; }
  if ((tid & 63) == 0) wmax[tid >> 6] = mbits;
  __syncthreads();
  if (tid == 0) {
    unsigned m = wmax[0];
    for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
    if (m) atomicMax(&amax[(blockIdx.x + blockIdx.y) & 63], m);
  }
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    if (slot[k] < 0) continue;
    const int z = z0 + k;
    float* dst = entry_slot(L, keys[slot[k]], off[slot[k]] + rank[k], 8);
    *reinterpret_cast<float4*>(dst) = make_float4(__int_as_float((z << 20) | (y << 10) | x), f[k][0], f[k][1], f[k][2]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(gv[k][0], gv[k][1], gv[k][2], 0.f);
  }
}

// The fixed point: unit u = 2^(E - FXBITS), E = the exponent of max |d_out| (max = m 2^E, m in [0.5, 1)); int64 sums hold 2^23
// terms of the largest size.  A contribution v w is converted through the 32-bit converter with a scale PER ENTRY (the 64-bit
// converter is a ~10-instruction sequence, and there are 64 conversions per entry: it made pass C VALU-bound):
//   e = the exponent of the entry's own max |d_out[c]|, D = E - e >= 0
//   D <= SHMAX - 1: x = rint(v w 2^(29 - e)) (|x| <= 2^29: 29 bits below the ENTRY's maximum, finer than fp32's 24), shifted left
//                   by sh = SHMAX - D >= 1 as two 32-bit shifts (lo = x << sh, hi = x >> (32 - sh), arithmetic)
//   D >= SHMAX:     the entry is small against the tensor: x = rint(v w 2^(FXBITS - 1 - E)) (< 2^28), sh = 1 -- 2^-39 of the
//                   tensor's maximum per unit, where round 5's single scale stopped at 2^-30
constexpr int SHMAX = FXBITS - 29;
__device__ __forceinline__ bool fx_global(const unsigned* __restrict__ amax, int& E, float& inv) {
  unsigned m = amax[threadIdx.x & 63];
  for (int o = 32; o; o >>= 1) { const unsigned v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
  const bool finite = m < 0x7f800000u;
  E = 0;
  (void)frexpf(m && finite ? __uint_as_float(m) : 1.f, &E);
  inv = ldexpf(1.f, E - FXBITS);
  return finite;
}
__device__ __forceinline__ void fx_entry(unsigned vb /* bits of the entry's max |d_out[c]|, != 0 */, int E, float& fs, int& sh) {
  int e = 0;
  (void)frexpf(__uint_as_float(vb), &e);
  const int D = E - e;
  const bool big = D <= SHMAX - 1;
  sh = big ? SHMAX - D : 1;
  fs = ldexpf(1.f, big ? 29 - e : FXBITS - 1 - E);
}
__device__ __forceinline__ unsigned long long fx_make(float x, int sh) {
  int xi;                                                     // floor(x + 0.5) in ONE instruction (round-to-nearest-even is two: v_rndne + v_cvt)
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(xi) : "v"(x));
  return ((unsigned long long)(unsigned)(xi >> (32 - sh)) << 32) | (unsigned)(xi << sh);
}

__device__ __forceinline__ int border_index(int lz, int ly, int lx) {      // cells with max(l) == 8
  if (lz == 8) return ly * 9 + lx;                   // 81
  if (ly == 8) return 81 + lz * 9 + lx;              // 72 (lz < 8)
  return 153 + lz * 8 + ly;                          // 64 (lx == 8, lz, ly < 8)
}

// ---- C: the tile's payload -> fixed-point window -> owned cells (plain stores) + high-face cells (side buffer).  One workgroup
// per (tile, group of eight channels); ONE lane per entry and all eight channels: the corner / weight arithmetic of an entry is
// done once.  46 KB of LDS: three workgroups per CU.  What the counters say (level 1, 160x192x160, profiles/r06*_accumulate*):
// the CU's VALU + LDS pipes are the bound, not latency -- 314 M ds_add_u64 at ~10 LDS cycles per wave instruction (a wave's 64
// lanes hit the 32 64-bit banks ~3.5 deep: the lists are in source order, random in the tile's cells) = ~80 us, the 5 VALU
// per contribution (multiply, round, convert, two shifts) ~60 us, and they overlap badly; one / two / three workgroups per CU
// run 356 / 277 / 254 us.  Measured and NOT kept: (a) counting-sorting each chunk by base cell so that a wave walks consecutive
// banks -- same-address duplicates replace the bank conflicts and the sort's barriers cost 90 us; (b) persistent workgroups with
// the next item's payload prefetched -- no faster (the payload's round trip is hidden by the other workgroups already) and
// the static striding unbalances (254 against 236 us).
struct AccEntry { float4 hd, ga, gb; };
template <int NCH>
__device__ __forceinline__ void acc_load(const float* __restrict__ lp, unsigned i, unsigned n, int S, int c0, AccEntry& a) {
  const float* ep = lp + (size_t)(i < n ? i : 0u) * S;       // (n > 0 here: entry 0 exists; lanes past the end are masked at use)
  a.hd = *reinterpret_cast<const float4*>(ep);
  a.ga = *reinterpret_cast<const float4*>(ep + 4 + c0);
  if constexpr (NCH == 8) a.gb = *reinterpret_cast<const float4*>(ep + 8 + c0);
  else a.gb = make_float4(0.f, 0.f, 0.f, 0.f);              // (NCH 4: a half group; NCH 3: three channels + a zero)
}
template <int NCH>
__device__ __forceinline__ void acc_entry(const AccEntry& a, unsigned long long* win, const Geo& g, int oz, int oy, int ox, int E, bool finite) {
  const float gvv[8] = {a.ga.x, a.ga.y, a.ga.z, a.ga.w, a.gb.x, a.gb.y, a.gb.z, a.gb.w};
  unsigned vb = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) { const unsigned u = __float_as_uint(fabsf(gvv[c])); vb = u > vb ? u : vb; }
  if (!vb) return;                                            // (C > 8: this group of the entry is all zero)
  float fs;
  int sh;
  fx_entry(finite ? vb : 0x3f800000u, E, fs, sh);
  const int pk = __float_as_int(a.hd.x);
  Entry e;
  (void)make_entry(a.hd.y, a.hd.z, a.hd.w, pk >> 20, (pk >> 10) & 1023, pk & 1023, g, e);
  float vs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) vs[c] = gvv[c] * fs;
  const int cell0 = ((e.bz - oz) * WN + (e.by - oy)) * WN + (e.bx - ox);
  // every corner inside the volume (all but the entries at the volume's faces): no per-corner tests
  const bool inner = e.bz >= 0 && e.bz + 1 < g.D && e.by >= 0 && e.by + 1 < g.H && e.bx >= 0 && e.bx + 1 < g.W;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
    if (!inner) {
      const int gz = e.bz + dz, gy = e.by + dy, gx = e.bx + dx;
      if (gz < 0 || gz >= g.D || gy < 0 || gy >= g.H || gx < 0 || gx >= g.W) continue;
    }
    const float w = (dz ? e.fz : 1.f - e.fz) * (dy ? e.fy : 1.f - e.fy) * (dx ? e.fx : 1.f - e.fx);
    unsigned long long* wp = win + (cell0 + (dz * WN + dy) * WN + dx);       // window = [channel][cell]
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const unsigned long long cv = fx_make(vs[k] * w, sh);
#ifdef MODET_TUNING
      if (g.dbg & 16) { wp[k * CELLS] = cv; continue; }                      // plain store instead of the atomic
      if (g.dbg & 32) { if (cv == 0x123456789abcull) wp[k * CELLS] = cv; continue; }   // no LDS operation at all
#endif
      atomicAdd(wp + k * CELLS, cv);
    }
  }
}

// NCH = channels of a work item: 8, 4 (a half group: 23 KB of LDS, six 256-lane workgroups per CU) or 3 (C == 3: payload entries of
// 8 words, d_src / side-buffer cells of 3 / 4 floats); NT = lanes of the workgroup
template <int NCH, int NT>
__global__ __launch_bounds__(NT) void accumulate_kernel(const Lists L, const unsigned* __restrict__ amax,
                                                        float* __restrict__ dsrc, float* __restrict__ border, const Geo g) {
  __shared__ __attribute__((aligned(16))) unsigned long long win[CELLS * NCH + 1];
  const int tid = threadIdx.x;
  const int C = g.C, S = NCH == 3 ? 8 : 4 + C, BC = NCH == 3 ? 4 : C;
  const int tile = blockIdx.x, c0 = blockIdx.y * (NCH == 3 ? 8 : NCH);           // tile: over all samples
  const int b = tile / g.ntiles;
  int t = tile - b * g.ntiles;
  const int ox = (t % g.tx) * TL; t /= g.tx;
  const int oy = (t % g.ty) * TL;
  const int oz = (t / g.ty) * TL;
  const unsigned handed = L.cursor[tile], n = handed < CAP ? handed : CAP;
  dsrc += (int64_t)b * g.D * g.H * g.W * C + c0;
  if (n == 0) {                                               // nothing lands here: zeros, and no side-buffer cells (the border pass tests the count)
    for (int cell = tid; cell < TL * TL * TL; cell += NT) {
      const int lx = cell & 7, ly = (cell >> 3) & 7, lz = cell >> 6;
      const int gz = oz + lz, gy = oy + ly, gx = ox + lx;
      if (gz < g.D && gy < g.H && gx < g.W) {
        float* dst = dsrc + (((int64_t)gz * g.H + gy) * g.W + gx) * C;
        if constexpr (NCH == 3) { dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f; }
        else {
          *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (NCH == 8) *reinterpret_cast<float4*>(dst + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    return;
  }
  int E;
  float inv_scale;
  const bool finite = fx_global(amax, E, inv_scale);
  {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2* w2 = reinterpret_cast<u64x2*>(win);
    for (int i = tid; i < (CELLS * NCH + 1) / 2; i += NT) w2[i] = (u64x2){0ull, 0ull};
  }
  __syncthreads();
  const float* lp = L.list + (size_t)tile * CAP * S;
  for (unsigned i = tid; i < n; i += NT) {
    AccEntry a;
    acc_load<NCH>(lp, i, n, S, c0, a);
    acc_entry<NCH>(a, win, g, oz, oy, ox, E, finite);
  }
  if (handed > CAP) {                                         // (uniform, rare: a folded tile) its entries beyond CAP are in the overflow list
    const unsigned novf = *L.ovf_count;
    for (unsigned i = tid; i < novf; i += NT) {
      if (L.ovf_tile[i] != (unsigned)tile) continue;
      AccEntry a;
      acc_load<NCH>(L.ovf_list, i, novf, S, c0, a);
      acc_entry<NCH>(a, win, g, oz, oy, ox, E, finite);
    }
  }
  __syncthreads();
  // flush: one cell (NCH channels) per thread and trip
  const float poison = finite ? 0.f : __uint_as_float(0x7fc00000u);
  for (int cell = tid; cell < CELLS; cell += NT) {
    const int lx = cell % WN, lr = cell / WN, ly = lr % WN, lz = lr / WN;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k) o[k] = (float)(long long)win[k * CELLS + cell] * inv_scale + poison;
    const bool owned = lz < TL && ly < TL && lx < TL;
    float* dst;
    if (owned) {
      const int gz = oz + lz, gy = oy + ly, gx = ox + lx;
      if (gz >= g.D || gy >= g.H || gx >= g.W) continue;
      dst = dsrc + (((int64_t)gz * g.H + gy) * g.W + gx) * C;
    } else {
      dst = border + ((int64_t)tile * NBORDER + border_index(lz, ly, lx)) * BC + c0;
    }
    if (NCH != 3 || !owned) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    else { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; }
    if constexpr (NCH == 8) *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// ---- D: owned cells on a low face of their tile (169 of 512) += the high-face cells of the up to seven neighbours that have
// entries; every side-buffer cell is read exactly once, in a fixed order
template <bool C3>
__global__ __launch_bounds__(256) void border_kernel(float* __restrict__ dsrc, const float* __restrict__ border,
                                                     const unsigned* __restrict__ cursor, const Geo g) {
  const int tile = blockIdx.x;
  const int b = tile / g.ntiles;
  int t = tile - b * g.ntiles;
  const int tx = t % g.tx; t /= g.tx;
  const int ty = t % g.ty, tz = t / g.ty;
  const int C = g.C, q4 = C3 ? 1 : 2, cb = blockIdx.y * 8, BC = C3 ? 4 : C;      // a thread = four channels of one face cell (C == 3: the cell); blockIdx.y = eight channels
  // which of the seven lower neighbours exist and have entries (uniform over the workgroup)
  unsigned live = 0;
#pragma unroll
  for (int m = 1; m < 8; ++m) {
    const int dz = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
    if ((dz && tz == 0) || (dy && ty == 0) || (dx && tx == 0)) continue;
    const int nt = b * g.ntiles + ((tz - dz) * g.ty + (ty - dy)) * g.tx + (tx - dx);
    if (cursor[nt] != 0u) live |= 1u << m;
  }
  if (!live) return;
  dsrc += (int64_t)b * g.D * g.H * g.W * C;
  for (int j = threadIdx.x; j < 169 * q4; j += 256) {
    const int ch = cb + (j % q4) * 4, k = j / q4;
    int lz, ly, lx;
    if (k < 64) { lz = 0; ly = k >> 3; lx = k & 7; }
    else if (k < 120) { const int r = k - 64; ly = 0; lz = 1 + r / 8; lx = r & 7; }
    else { const int r = k - 120; lx = 0; lz = 1 + r / 7; ly = 1 + r % 7; }
    const int gz = tz * 8 + lz, gy = ty * 8 + ly, gx = tx * 8 + lx;
    if (gz >= g.D || gy >= g.H || gx >= g.W) continue;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
#pragma unroll
    for (int m = 1; m < 8; ++m) {
      const int dz = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
      if (!(live & (1u << m)) || (dz && lz) || (dy && ly) || (dx && lx)) continue;
      const int nt = b * g.ntiles + ((tz - dz) * g.ty + (ty - dy)) * g.tx + (tx - dx);
      const float4 v = *reinterpret_cast<const float4*>(border + ((int64_t)nt * NBORDER + border_index(dz ? 8 : lz, dy ? 8 : ly, dx ? 8 : lx)) * BC + ch);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      any = true;
    }
    if (!any) continue;
    float* dq = dsrc + (((int64_t)gz * g.H + gy) * g.W + gx) * C + ch;
    if constexpr (C3) {
      dq[0] += s.x; dq[1] += s.y; dq[2] += s.z;
    } else {
      float4* dp = reinterpret_cast<float4*>(dq);
      float4 d = *dp;
      d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
      *dp = d;
    }
  }
}

struct Ws { unsigned *amax, *ovf_count, *cursor, *ovf_tile; float *list, *ovf_list, *border; size_t bytes; };
// [amax 64][overflow count 4][cursor nt] unsigned -- zeroed by the launcher -- [overflow tile ids B*D*H*W] unsigned (padded to 16 bytes),
// [tile lists nt x CAP x S] float, [overflow list B*D*H*W x S] float, [side buffer nt x 217 x BC] float
// (nt = tiles of all samples; S = 4 + C words per entry, C == 3: S = 8 and side-buffer cells of BC = 4)
__host__ bool ws_layout(void* ws, int B, int D, int H, int W, int C, Ws& w) {
  if (B < 1 || D < 1 || H < 1 || W < 1 || (C != 3 && (C < 8 || C % 8 != 0)) || D > 1024 || H > 1024 || W > 1024) return false;
  const size_t nt = (size_t)B * cdiv(D, 8) * cdiv(H, 8) * cdiv(W, 8);
  const size_t vox = (size_t)B * D * H * W;
  if (vox >= (1ull << 31) || nt >= (1u << 30)) return false;      // (32-bit entry indices)
  const size_t hdr = (68 + nt + vox + 3) / 4 * 4;
  const size_t S = C == 3 ? 8 : 4 + C, BC = C == 3 ? 4 : C;
  w.amax = (unsigned*)ws; w.ovf_count = w.amax + 64; w.cursor = w.ovf_count + 4; w.ovf_tile = w.cursor + nt;
  w.list = (float*)(w.amax + hdr);
  w.ovf_list = w.list + nt * CAP * S;
  w.border = w.ovf_list + vox * S;
  w.bytes = (hdr + nt * CAP * S + vox * S + nt * NBORDER * BC) * 4 + 256;
  return true;
}

int tiles_launch(const void* src, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow, const float* d_flow_add,
                 void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, int add_flow, modet_stream_t stream) {
  MODET_CHECK_PTR(flow); MODET_CHECK_PTR(d_out); MODET_CHECK_PTR(d_src); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  Ws w;
  if (!ws_layout(ws, B, D, H, W, C, w) || (C == 3 && src_bf16) || (add_flow && C != 3)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < w.bytes) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int dbg = 0;                 // tuning builds only: WT_DBG = bit mask of parts to leave out (timing experiments, wrong results)
#ifdef MODET_TUNING
  if (const char* e = getenv("WT_DBG")) dbg = atoi(e);
#endif
  Geo g{D, H, W, C, cdiv(D, 8), cdiv(H, 8), cdiv(W, 8), 0, B, dbg};
  g.ntiles = g.tz * g.ty * g.tx;
  const int nt = B * g.ntiles;
  const bool small = (int64_t)B * D * H * W < 400000;      // (levels 3-5; level 2, 614 k voxels, is faster with SZ = 4)
  const int bx_n = cdiv(W, SX), by_n = cdiv(H, SY), bz_n = cdiv(D, small ? 1 : 4);
  const dim3 bgrid(bx_n * by_n * bz_n, B);
  const Lists L{w.list, w.cursor, w.ovf_count, w.ovf_tile, w.ovf_list};
  modet_zero_async(w.amax, (size_t)(68 + nt) * 4, s);
  if (C == 3) {
    if (small) hipLaunchKernelGGL(fill_c3_kernel<1>, bgrid, dim3(256), 0, s, (const float*)src, flow, d_out, L, w.amax, d_flow,
                                  d_flow_add, g, bx_n, by_n, add_flow);
    else hipLaunchKernelGGL(fill_c3_kernel<4>, bgrid, dim3(256), 0, s, (const float*)src, flow, d_out, L, w.amax, d_flow,
                            d_flow_add, g, bx_n, by_n, add_flow);
    hipLaunchKernelGGL((accumulate_kernel<3, ACC>), dim3(nt, 1), dim3(ACC), 0, s, L, (const unsigned*)w.amax, d_src, w.border, g);
    hipLaunchKernelGGL(border_kernel<true>, dim3(nt, 1), dim3(256), 0, s, d_src, (const float*)w.border, (const unsigned*)w.cursor, g);
    return modet_launch_status();
  }
#define WT_FILL(S16, SZ) hipLaunchKernelGGL((fill_kernel<S16, SZ>), bgrid, dim3(256), 0, s, src, flow, d_out, L, w.amax, d_flow, \
                                            d_flow_add, g, bx_n, by_n, dbg)
  if (src_bf16) { if (small) WT_FILL(true, 1); else WT_FILL(true, 4); }
  else { if (small) WT_FILL(false, 1); else WT_FILL(false, 4); }
#undef WT_FILL
  int half = 0;
#ifdef MODET_TUNING
  if (const char* e = getenv("WT_NCH4")) half = atoi(e);
#endif
  if (half == 1) hipLaunchKernelGGL((accumulate_kernel<4, 256>), dim3(nt, C / 4), dim3(256), 0, s, L, (const unsigned*)w.amax, d_src, w.border, g);
  else if (half == 2) hipLaunchKernelGGL((accumulate_kernel<4, 512>), dim3(nt, C / 4), dim3(512), 0, s, L, (const unsigned*)w.amax, d_src, w.border, g);
  else if (half == 3) hipLaunchKernelGGL((accumulate_kernel<8, 256>), dim3(nt, C / 8), dim3(256), 0, s, L, (const unsigned*)w.amax, d_src, w.border, g);
  else hipLaunchKernelGGL((accumulate_kernel<8, ACC>), dim3(nt, C / 8), dim3(ACC), 0, s, L, (const unsigned*)w.amax, d_src, w.border, g);
  hipLaunchKernelGGL(border_kernel<false>, dim3(nt, C / 8), dim3(256), 0, s, d_src, (const float*)w.border, (const unsigned*)w.cursor, g);
  return modet_launch_status();
}
}  // namespace

extern "C" {

size_t modet_warp_bwd_dsrc_tiles_ws_bytes(int B, int D, int H, int W, int C) {
  Ws w;
  return ws_layout(nullptr, B, D, H, W, C, w) ? w.bytes : 0;
}

int modet_warp_bwd_dsrc_tiles(const float* flow, const float* d_out, float* d_src, void* ws, size_t ws_bytes, int B, int D, int H,
                              int W, int C, modet_stream_t stream) {
  return tiles_launch(nullptr, 0, flow, d_out, d_src, nullptr, nullptr, ws, ws_bytes, B, D, H, W, C, 0, stream);
}

int modet_warp_bwd_tiles(const void* src, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow,
                         const float* d_flow_add, void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, int add_flow,
                         modet_stream_t stream) {
  MODET_CHECK_PTR(src); MODET_CHECK_PTR(d_flow);
  return tiles_launch(src, src_bf16, flow, d_out, d_src, d_flow, d_flow_add, ws, ws_bytes, B, D, H, W, C, add_flow, stream);
}

}  // extern "C"
