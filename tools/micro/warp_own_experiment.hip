// EXPERIMENT RECORD -- not built, not part of libmodet_hip.so.  Measured on MI355X (profiles/r04e_warp_own_experiment.log)
// and NOT adopted: on the model's own flows (C = 8, 160x192x160) the source-owned form below takes 2.0-2.4 ms against 0.89 ms
// for warp_bwd2_kernel's L2 atomics.  Counters: 3.3 block visits per (voxel, pass); 94 % of the visits have two voxels with
// the same base corner (the randomly initialised model's flow is rough at voxel scale), so the per-corner election runs 2.4
// rounds; skipping the LDS accumulation leaves 0.82 ms, skipping d_flow as well 0.46 ms (scan + loads + stores alone).  The
// first form (256 threads sharing a tile through ds_add_f32) took 2.36 ms: LDS float atomics retire ~1 wave instruction per
// 100+ clocks.  Kept as the record of why warp backward stays on L2 atomics.
// SpatialTransformer backward, SOURCE-OWNED form: d_src without global atomics and without the zero fill.
//   reference call site: F.grid_sample(..., align_corners=True, mode='bilinear') backward in SpatialTransformer,
//   ModeT/models.py:25-67 (arithmetic = ATen grid_sampler_3d_backward: 8 corner weights scattered into d_src).
// warp.hip's kernels own OUTPUT voxels and scatter into d_src with L2 float atomics: ~3 lane-atomics per (voxel, channel)
// after the x / y / z merges on the model's flows, irregular, ~145 G/s -- that is their whole time (0.80 ms for the level-1
// features, C = 8, 160x192x160).  Here a workgroup owns a TILE OF d_src (S = SZ x SY x SX cells), accumulates it in LDS and
// writes it once with plain 16-byte stores:
//   * a table pass (warp_own_table_kernel) records, for every block of 2 x 4 x 8 output voxels (one wave), the bounding box
//     of the cells its sample points touch, and the union over super-blocks of 4 x 4 x 4 blocks (16 bytes per entry);
//   * the owner scans the super-block table (a lane per entry), then the 64 block entries of each hit (a lane per block);
//     the four waves share the hits round-robin.  A hit block is processed by one wave, a lane per voxel: trilinear
//     weights, and for each of the 8 corners that falls inside S an LDS float add per channel.  Any flow is handled
//     exactly -- a rough flow only lengthens the hit list;
//   * d_flow of an output voxel is computed by the workgroup whose tile holds the voxel's (clamped) base corner, from
//     global reads of the 8 corners -- every voxel has exactly one such owner, so d_flow is a plain store as well.
// Cost: flow / d_out of a block are re-read by each of the ~2-3 tiles its footprint touches (L2 hits: neighbouring tiles
// run on the same XCD), against ~3 L2 atomics per (voxel, channel) saved.
#include "common.h"

namespace {

constexpr int NTHR = 256;
constexpr int OBZ = 2, OBY = 4, OBX = 8;               // output block = one wave, a lane per voxel
constexpr int SUP = 4;                                 // super-block = SUP^3 blocks = 8 x 16 x 32 voxels
constexpr int SUPB = SUP * SUP * SUP;
constexpr int LIST_MAX = 512;                          // hit blocks of a tile held in LDS

struct Geo { int B, D, H, W, C, nsz, nsy, nsx; };      // nsz/y/x: super-blocks per dimension

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
// base corner of a sample coordinate, as warp.hip's tri_setup: floor, clamped to [-2, dim] (both ends have no cell in the volume)
__device__ __forceinline__ int base_of(float v, float& frac, int dim) {
  const float f = floorf(v);
  frac = v - f;
  return (int)fminf(fmaxf(f, -2.f), (float)dim);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// entry = {lo_z | lo_y << 16, lo_x | hi_z << 16, hi_y | hi_x << 16, 0}: CLAMPED base corners (to [0, dim - 1]); an empty
// block has lo = 0x7fff, hi = 0.  A block can matter to the tile [slo, shi] only if lo <= shi and hi + 1 >= slo in every
// dimension (its sample points touch cells base, base + 1; the d_flow owner is the tile holding the clamped base).
__device__ __forceinline__ uint4 pack_entry(int lz, int ly, int lx, int hz, int hy, int hx) {
  return make_uint4((unsigned)lz | ((unsigned)ly << 16), (unsigned)lx | ((unsigned)hz << 16), (unsigned)hy | ((unsigned)hx << 16), 0u);
}
struct TileBox { int lz, ly, lx, hz, hy, hx; };
__device__ __forceinline__ bool entry_hits(const uint4 e, const TileBox& t) {
  const int lz = e.x & 0xffff, ly = e.x >> 16, lx = e.y & 0xffff, hz = e.y >> 16, hy = e.z & 0xffff, hx = e.z >> 16;
  return lz <= t.hz && hz + 1 >= t.lz && ly <= t.hy && hy + 1 >= t.ly && lx <= t.hx && hx + 1 >= t.lx;
}

// one workgroup per (sample, super-block): wave w covers the blocks 16 w .. 16 w + 15 of the super-block
__global__ __launch_bounds__(NTHR) void warp_own_table_kernel(const float* __restrict__ flow, uint4* __restrict__ blk,
                                                             uint4* __restrict__ sup, const Geo g) {
  __shared__ int sm[4][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int s = blockIdx.x;
  const int ssx = s % g.nsx; s /= g.nsx;
  const int ssy = s % g.nsy; s /= g.nsy;
  const int ssz = s % g.nsz;
  const int b = s / g.nsz;
  const int vz = lane >> 5, vy = (lane >> 3) & 3, vx = lane & 7;
  int slz = 0x7fff, sly = 0x7fff, slx = 0x7fff, shz = 0, shy = 0, shx = 0;
  // the 16 blocks' flow loads first (independent), then the reductions
  float f[16][3];
  bool ok[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int jb = wave * 16 + i;
    const int pz = ((ssz * SUP + (jb >> 4)) * OBZ) + vz, py = ((ssy * SUP + ((jb >> 2) & 3)) * OBY) + vy, px = ((ssx * SUP + (jb & 3)) * OBX) + vx;
    ok[i] = pz < g.D && py < g.H && px < g.W;
    const int64_t n = ((int64_t)(b * g.D + (ok[i] ? pz : 0)) * g.H + (ok[i] ? py : 0)) * g.W + (ok[i] ? px : 0);
    f[i][0] = flow[n * 3]; f[i][1] = flow[n * 3 + 1]; f[i][2] = flow[n * 3 + 2];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int jb = wave * 16 + i;
    const int pz = ((ssz * SUP + (jb >> 4)) * OBZ) + vz, py = ((ssy * SUP + ((jb >> 2) & 3)) * OBY) + vy, px = ((ssx * SUP + (jb & 3)) * OBX) + vx;
    float fr;
    const int bz = clampi(base_of((float)pz + f[i][0], fr, g.D), 0, g.D - 1);
    const int by = clampi(base_of((float)py + f[i][1], fr, g.H), 0, g.H - 1);
    const int bx = clampi(base_of((float)px + f[i][2], fr, g.W), 0, g.W - 1);
    const int lz = wave_min_i(ok[i] ? bz : 0x7fff), ly = wave_min_i(ok[i] ? by : 0x7fff), lx = wave_min_i(ok[i] ? bx : 0x7fff);
    const int hz = wave_max_i(ok[i] ? bz : 0), hy = wave_max_i(ok[i] ? by : 0), hx = wave_max_i(ok[i] ? bx : 0);
    if (lane == 0) blk[((int64_t)blockIdx.x * SUPB) + jb] = pack_entry(lz, ly, lx, hz, hy, hx);
    slz = min(slz, lz); sly = min(sly, ly); slx = min(slx, lx);
    shz = max(shz, hz); shy = max(shy, hy); shx = max(shx, hx);
  }
  if (lane == 0) { sm[wave][0] = slz; sm[wave][1] = sly; sm[wave][2] = slx; sm[wave][3] = shz; sm[wave][4] = shy; sm[wave][5] = shx; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      slz = min(slz, sm[w][0]); sly = min(sly, sm[w][1]); slx = min(slx, sm[w][2]);
      shz = max(shz, sm[w][3]); shy = max(shy, sm[w][4]); shx = max(shx, sm[w][5]);
    }
    sup[blockIdx.x] = pack_entry(slz, sly, slx, shz, shy, shx);
  }
}

template <int N> __device__ __forceinline__ void ldv_own(const float* p, float (&r)[N]) {
  if constexpr (N == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = p[i];
  }
}
template <int N> __device__ __forceinline__ void stv_own(float* p, const float (&r)[N]) {
  if constexpr (N == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = r[i];
  }
}

#ifdef MODET_TUNING
__device__ unsigned long long g_own_stats[8];         // visits, non-distinct visits, election rounds, corner instr executed, owner visits
#define OWN_STAT(i, v) do { if (lane == 0) atomicAdd(&g_own_stats[i], (unsigned long long)(v)); } while (0)
#else
#define OWN_STAT(i, v) do { } while (0)
#endif

struct OwnArgs {
  const float* src; const float* flow; const float* dout; float* dsrc; float* dflow;
  const uint4* blk; const uint4* sup;
  Geo g;
  int dbg;                                              // tuning builds: bit 0 skip the LDS accumulation, bit 1 skip d_flow, bit 2 skip both loads
  int tz, ty, tx, ntiles, per_xcd;                      // d_src tiles per dimension, per sample; blockIdx -> tile chunks per XCD
};

// One WAVE per workgroup and per tile: CG channels of d_src per pass (1 | 4), S = SZ x SY x SX cells, acc[cell][CG] + a byte
// of tag per cell in LDS.  LDS float atomics are far too slow here (ds_add_f32 measured at ~100+ clocks per wave
// instruction: the 256-thread / shared-tile form of this kernel took 2.4 ms); instead the lanes of one corner instruction
// elect a winner per cell -- write the lane id to tag[cell], read it back, the lane that reads its own id adds with a plain
// 16-byte read-modify-write, the others go round again -- which is exact for any flow and one round when no two voxels of
// the block share a cell (the common case).  LDS instructions of a wave execute in order, so no barrier is involved.
template <int CG, int SZ, int SY, int SX>
__global__ __launch_bounds__(64) void warp_own_kernel(const OwnArgs a) {
  constexpr int CELLS = SZ * SY * SX;
  __shared__ __attribute__((aligned(16))) float acc[CELLS * CG];
  __shared__ unsigned char tagm[CELLS];
  __shared__ unsigned char hmapm[512];
  __shared__ int listm[LIST_MAX];
  volatile unsigned char* tag = tagm;
  volatile unsigned char* hmap = hmapm;
  volatile int* list = listm;
  const Geo g = a.g;
  const int D = g.D, H = g.H, W = g.W, C = g.C;
  const int lane = threadIdx.x;
  // consecutive tiles on one XCD (blockIdx round-robins the 8 XCDs): neighbours re-read the same blocks from that L2
  const int til = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (til >= a.ntiles) return;
  int t = til;
  const int ix = t % a.tx; t /= a.tx;
  const int iy = t % a.ty; t /= a.ty;
  const int iz = t % a.tz;
  const int b = t / a.tz;
  TileBox tb;
  tb.lz = iz * SZ; tb.ly = iy * SY; tb.lx = ix * SX;
  tb.hz = min(tb.lz + SZ, D) - 1; tb.hy = min(tb.ly + SY, H) - 1; tb.hx = min(tb.lx + SX, W) - 1;
  const int ez = tb.hz - tb.lz + 1, ey = tb.hy - tb.ly + 1, ex = tb.hx - tb.lx + 1;
  const int nsup = g.nsz * g.nsy * g.nsx;
  const uint4* supb = a.sup + (int64_t)b * nsup;
  const uint4* blkb = a.blk + (int64_t)b * nsup * SUPB;
  const int vz = lane >> 5, vy = (lane >> 3) & 3, vx = lane & 7;
  const int64_t sampleV = (int64_t)D * H * W;
  const float* srcb = a.src + (int64_t)b * sampleV * C;
  const int npass = C / CG;

  // hit list: (super-block index << 6) | block, wave-uniform entries; built by the first pass and re-used by the later ones
  // unless it overflowed (rough flow: processed in chunks, every pass scans again)
  int nl = 0;
  bool list_complete = false;

  // ---- process list[0 .. n): a lane per voxel of each block; the next block's flow / d_out are in flight meanwhile
  auto process = [&](const int n_ent, const int pass) {
    if (n_ent == 0) return;
    float nf[3], ng[CG];
    int64_t nn;
    bool ninv;
    int npz, npy, npx;
    auto issue = [&](int i) {
      const int e = __builtin_amdgcn_readfirstlane(list[i]);
      const int sidx = e >> 6, jb = e & 63;
      const int ssx = sidx % g.nsx, ssy = (sidx / g.nsx) % g.nsy, ssz = sidx / (g.nsx * g.nsy);
      npz = (ssz * SUP + (jb >> 4)) * OBZ + vz; npy = (ssy * SUP + ((jb >> 2) & 3)) * OBY + vy; npx = (ssx * SUP + (jb & 3)) * OBX + vx;
      ninv = npz < D && npy < H && npx < W;
      nn = (int64_t)b * sampleV + ((int64_t)(ninv ? npz : 0) * H + (ninv ? npy : 0)) * W + (ninv ? npx : 0);
      nf[0] = a.flow[nn * 3]; nf[1] = a.flow[nn * 3 + 1]; nf[2] = a.flow[nn * 3 + 2];
      ldv_own(a.dout + nn * C + pass * CG, ng);
    };
    issue(0);
    for (int i = 0; i < n_ent; ++i) {
      const float f0 = nf[0], f1 = nf[1], f2 = nf[2];
      float gch[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) gch[c] = ng[c];
      const int64_t n = nn;
      const bool inv = ninv;
      const int pz = npz, py = npy, px = npx;
      if (i + 1 < n_ent) issue(i + 1);
      float fz, fy, fx;
      const int bz = base_of((float)pz + f0, fz, D), by = base_of((float)py + f1, fy, H), bx = base_of((float)px + f2, fx, W);
      const float wzv[2] = {1.f - fz, fz}, wyv[2] = {1.f - fy, fy}, wxv[2] = {1.f - fx, fx};
      const int cz0 = bz - tb.lz, cy0 = by - tb.ly, cx0 = bx - tb.lx;
      // no two voxels of the block share a base corner (one election on a hash of the base that is injective over the
      // span of a smooth block) => for every corner the lanes' cells are distinct: plain read-modify-writes
      const int hsh = ((bz & 3) << 7) | ((by & 7) << 4) | (bx & 15);
      if (inv) hmap[hsh] = (unsigned char)lane;
      const bool distinct = !__builtin_amdgcn_ballot_w64(inv && hmap[hsh] != (unsigned char)lane);
      OWN_STAT(0, 1); OWN_STAT(1, distinct ? 0 : 1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (a.dbg & 1) break;
        const int cz = cz0 + (q >> 2), cy = cy0 + ((q >> 1) & 1), cx = cx0 + (q & 1);
        bool pend = inv && (unsigned)cz < (unsigned)ez && (unsigned)cy < (unsigned)ey && (unsigned)cx < (unsigned)ex;
        const float wgt = wzv[q >> 2] * wyv[(q >> 1) & 1] * wxv[q & 1];
        const int cell = pend ? (cz * SY + cy) * SX + cx : 0;
        if (__builtin_amdgcn_ballot_w64(pend)) OWN_STAT(3, 1);
        if (distinct) {
          if (pend) {
            float v[CG];
            ldv_own(acc + cell * CG, v);
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = fmaf(wgt, gch[c], v[c]);
            stv_own(acc + cell * CG, v);
          }
        } else {
          while (__builtin_amdgcn_ballot_w64(pend)) {
            OWN_STAT(2, 1);
            if (pend) tag[cell] = (unsigned char)lane;
            const bool win = pend && tag[cell] == (unsigned char)lane;
            if (win) {
              float v[CG];
              ldv_own(acc + cell * CG, v);
#pragma unroll
              for (int c = 0; c < CG; ++c) v[c] = fmaf(wgt, gch[c], v[c]);
              stv_own(acc + cell * CG, v);
            }
            pend = pend && !win;
          }
        }
      }
      // ---- d_flow: the tile holding the clamped base corner owns the voxel (first pass only, all C channels)
      if (a.dflow && pass == 0 && !(a.dbg & 2)) {
        const int oz = clampi(bz, 0, D - 1), oy = clampi(by, 0, H - 1), ox = clampi(bx, 0, W - 1);
        const bool own = inv && oz >= tb.lz && oz <= tb.hz && oy >= tb.ly && oy <= tb.hy && ox >= tb.lx && ox <= tb.hx;
        if (__builtin_amdgcn_ballot_w64(own)) {
          OWN_STAT(4, 1);
          float dot[8];
          bool okc[8];
          int64_t off[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int zz = bz + (q >> 2), yy = by + ((q >> 1) & 1), xx = bx + (q & 1);
            okc[q] = own && zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
            off[q] = okc[q] ? (((int64_t)zz * H + yy) * W + xx) * C : 0;
            dot[q] = 0.f;
          }
          for (int c0 = 0; c0 < C; c0 += CG) {
            float gv[CG];
            if (c0 == 0) {
#pragma unroll
              for (int c = 0; c < CG; ++c) gv[c] = gch[c];
            } else {
              ldv_own(a.dout + n * C + c0, gv);
            }
            float sv[8][CG];
#pragma unroll
            for (int q = 0; q < 8; ++q) ldv_own(srcb + off[q] + c0, sv[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
              for (int c = 0; c < CG; ++c) dot[q] = fmaf(sv[q][c], gv[c], dot[q]);
          }
          float gz = 0.f, gy = 0.f, gx = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
            const float dq = okc[q] ? dot[q] : 0.f;
            gz += (dz ? 1.f : -1.f) * wyv[dy] * wxv[dx] * dq;
            gy += (dy ? 1.f : -1.f) * wzv[dz] * wxv[dx] * dq;
            gx += (dx ? 1.f : -1.f) * wzv[dz] * wyv[dy] * dq;
          }
          if (own) {
            float* dfp = a.dflow + n * 3;
            dfp[0] = gz; dfp[1] = gy; dfp[2] = gx;
          }
        }
      }
    }
  };

  for (int pass = 0; pass < npass; ++pass) {
    for (int i = lane; i < CELLS * CG; i += 64) acc[i] = 0.f;
    if (list_complete) {
      process(nl, pass);
    } else {
      bool flushed = false;
      nl = 0;
      for (int s0 = 0; s0 < nsup; s0 += 64) {
        const int si = s0 + lane;
        const uint4 se = supb[min(si, nsup - 1)];
        unsigned long long sm = __builtin_amdgcn_ballot_w64(si < nsup && entry_hits(se, tb));
        while (sm) {
          // up to four hit super-blocks' block entries in flight at once
          int sj[4];
          uint4 be[4];
          int ns = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sj[k] = sm ? __builtin_ctzll(sm) : -1;
            if (sm) { sm &= sm - 1; ++ns; }
            be[k] = blkb[(int64_t)(s0 + max(sj[k], 0)) * SUPB + lane];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < ns) {
              const bool hit = entry_hits(be[k], tb);
              const unsigned long long bm = __builtin_amdgcn_ballot_w64(hit);
              const int pos = nl + __builtin_popcountll(bm & ((1ull << lane) - 1ull));
              if (hit) list[pos] = ((s0 + sj[k]) << 6) | lane;
              nl += __builtin_popcountll(bm);
              if (nl > LIST_MAX - 64) { process(nl, pass); nl = 0; flushed = true; }
            }
          }
        }
      }
      process(nl, pass);
      list_complete = !flushed;
    }
    // ---- the tile is complete: plain stores, x fastest
    for (int i = lane; i < CELLS; i += 64) {
      const int cx = i % SX, cy = (i / SX) % SY, cz = i / (SX * SY);
      if (cz < ez && cy < ey && cx < ex) {
        float v[CG];
        ldv_own(acc + i * CG, v);
        float* dp = a.dsrc + ((int64_t)b * sampleV + ((int64_t)(tb.lz + cz) * H + tb.ly + cy) * W + tb.lx + cx) * C + pass * CG;
        stv_own(dp, v);
      }
    }
  }
}

struct OwnPlan { int nsz, nsy, nsx; size_t blk_bytes, sup_bytes; };
inline OwnPlan own_plan(int B, int D, int H, int W) {
  OwnPlan p;
  p.nsz = cdiv(cdiv(D, OBZ), SUP); p.nsy = cdiv(cdiv(H, OBY), SUP); p.nsx = cdiv(cdiv(W, OBX), SUP);
  const size_t nsup = (size_t)B * p.nsz * p.nsy * p.nsx;
  p.sup_bytes = nsup * 16;
  p.blk_bytes = nsup * SUPB * 16;
  return p;
}

template <int CG, int SZ, int SY, int SX>
void own_launch(OwnArgs a, hipStream_t s) {
  const Geo& g = a.g;
  a.tz = cdiv(g.D, SZ); a.ty = cdiv(g.H, SY); a.tx = cdiv(g.W, SX);
  a.ntiles = g.B * a.tz * a.ty * a.tx;
  a.per_xcd = cdiv(a.ntiles, 8);
  hipLaunchKernelGGL((warp_own_kernel<CG, SZ, SY, SX>), dim3(a.per_xcd * 8), dim3(64), 0, s, a);
}

}  // namespace

// ---- internal interface for warp.hip (modet_warp_bwd_ws)
// d_src AND optionally d_flow of a plain trilinear warp (no add_flow, no flow bound), C == 1 or C % 4 == 0
bool modetx_warp_own_eligible(int B, int D, int H, int W, int C) {
  if (modet_tuning_env("MODET_WARP_OWN") == '0') return false;
  if (!(C == 1 || (C % 4 == 0 && C <= 64))) return false;
  if (D > 32000 || H > 32000 || W > 32000) return false;
  return (int64_t)B * D * H * W >= 64 * 64 * 64;       // below: too few tiles to fill the chip, the atomics are cheap there
}
size_t modetx_warp_own_ws_bytes(int B, int D, int H, int W) {
  const OwnPlan p = own_plan(B, D, H, W);
  return p.blk_bytes + p.sup_bytes;
}
int modetx_warp_own_bwd(const float* src, const float* flow, const float* d_out, float* d_src, float* d_flow, void* ws, int B,
                        int D, int H, int W, int C, hipStream_t s) {
  const OwnPlan p = own_plan(B, D, H, W);
  OwnArgs a;
  a.src = src; a.flow = flow; a.dout = d_out; a.dsrc = d_src; a.dflow = d_flow;
  uint4* blk = reinterpret_cast<uint4*>(ws);
  uint4* sup = reinterpret_cast<uint4*>(reinterpret_cast<char*>(ws) + p.blk_bytes);
  a.blk = blk; a.sup = sup;
  a.g = Geo{B, D, H, W, C, p.nsz, p.nsy, p.nsx};
  hipLaunchKernelGGL(warp_own_table_kernel, dim3(B * p.nsz * p.nsy * p.nsx), dim3(NTHR), 0, s, flow, blk, sup, a.g);
  const char tv = modet_tuning_env("MODET_WARP_OWN_TILE");
  const char dv = modet_tuning_env("MODET_WARP_OWN_DBG");
  a.dbg = dv ? dv - '0' : 0;
  if (C == 1) {
    if (tv == 'b') own_launch<1, 8, 16, 32>(a, s);
    else own_launch<1, 8, 16, 16>(a, s);
  } else {
    if (tv == 'b') own_launch<4, 8, 16, 16>(a, s);
    else if (tv == 'c') own_launch<4, 4, 8, 16>(a, s);
    else own_launch<4, 8, 8, 16>(a, s);
  }
  return modet_launch_status();
}

#ifdef MODET_TUNING
extern "C" int modet_debug_warp_own_stats(unsigned long long* out, int reset) {      // not in the header: tuning builds only
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_own_stats), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_own_stats), z, sizeof(z)); }
  return 0;
}
#endif
