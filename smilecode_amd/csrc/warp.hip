// SpatialTransformer warp / flow composition, trilinear x2 upsample, layout changes, CWM tail and the
// label-warp + Dice-count evaluation tail.  All HBM-bound gathers/streams: channels-last so one 8-corner
// address computation serves every channel of a voxel, float4 per lane where C allows it.
//   reference: ModeT/models.py:25-67 (SpatialTransformer), :354/:257-261 (Upsample), :263-275 (CWM),
//              ModeT/utils.py:30-106 (nearest label warp, dice_val_VOI).
#include "common.h"

namespace {

constexpr int BLK = 256;

template <int CPT> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<3> { using T = float3; };   // loaded element-wise (12 B, no alignment guarantee)
template <> struct Vec<1> { using T = float; };

template <int CPT> __device__ __forceinline__ void ldv(const float* p, float (&r)[CPT]) {
  if constexpr (CPT == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else {
#pragma unroll
    for (int i = 0; i < CPT; ++i) r[i] = p[i];
  }
}
template <int CPT> __device__ __forceinline__ void stv(float* p, const float (&r)[CPT]) {
  if constexpr (CPT == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int i = 0; i < CPT; ++i) p[i] = r[i];
  }
}

struct Tri {            // trilinear footprint of one sample point
  int z0, y0, x0;
  float fz, fy, fx;
};
__device__ __forceinline__ Tri tri_setup(float z, float y, float x) {
  Tri t;
  const float zf = floorf(z), yf = floorf(y), xf = floorf(x);
  t.fz = z - zf; t.fy = y - yf; t.fx = x - xf;
  // clamp before the int conversion so wild flows cannot overflow; anything outside [-2, dim] is invalid anyway
  t.z0 = (int)fminf(fmaxf(zf, -2.f), 1.0e9f);
  t.y0 = (int)fminf(fmaxf(yf, -2.f), 1.0e9f);
  t.x0 = (int)fminf(fmaxf(xf, -2.f), 1.0e9f);
  return t;
}

// ------------------------------------------------------------------------------------------------ warp fwd
// O16 (BASELINE.json configs[4], bf16 storage of the warped features; CPT = 4): the output is rounded to bf16 (nearest even)
// S16: src holds bf16 (widened on load; CPT = 4).  Everything between the load and the store is fp32.
template <int CPT, bool S16>
__device__ __forceinline__ void ldsrc(const float* base, int64_t el, float (&r)[CPT]) {
  if constexpr (S16 && CPT == 4) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + el);
    r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
  } else {
    ldv<CPT>(base + el, r);
  }
}
template <int CPT, bool O16 = false, bool S16 = false>
__global__ __launch_bounds__(BLK) void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                       float* __restrict__ out, int D, int H, int W, int C, int G,
                                                       int64_t total, int mode, int add_flow) {
  const int64_t V = (int64_t)D * H * W;
  // 32-bit index arithmetic (the host guarantees total < 2^31): 64-bit div/mod are ~100-instruction software routines
  const unsigned utotal = (unsigned)total, uG = (unsigned)G, uW = (unsigned)W, uH = (unsigned)H, uD = (unsigned)D;
  for (unsigned idx = blockIdx.x * BLK + threadIdx.x; idx < utotal; idx += gridDim.x * BLK) {
    const int g = (int)(idx % uG);
    const unsigned nn = idx / uG;              // b*V + voxel
    const int xi = (int)(nn % uW);
    const unsigned t2 = nn / uW;
    const int yi = (int)(t2 % uH);
    const unsigned t3 = t2 / uH;
    const int zi = (int)(t3 % uD);
    const int64_t b = t3 / uD, n = nn;
    const float* fp = flow + n * 3;
    const float f0 = fp[0], f1 = fp[1], f2 = fp[2];
    const float z = (float)zi + f0, y = (float)yi + f1, x = (float)xi + f2;
    const int64_t sbe = b * V * C + g * CPT;       // element offset of (sample, channel group) inside src
    float acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
    if (mode == 1) {
      const float zr = rintf(z), yr = rintf(y), xr = rintf(x);     // round half to even = nearbyint
      if (zr >= 0.f && zr < (float)D && yr >= 0.f && yr < (float)H && xr >= 0.f && xr < (float)W)
        ldsrc<CPT, S16>(src, sbe + (((int64_t)zr * H + (int64_t)yr) * W + (int64_t)xr) * C, acc);
    } else {
      // branch-free: the eight corner loads are issued back to back (out-of-range corners read a clamped, valid voxel and
      // the VALUE is replaced by 0 afterwards -- as grid_sample's zeros padding and the backward kernels do, so an Inf / NaN
      // border voxel cannot leak in through 0 * Inf).  A load per `if (inside)` compiled to eight load -> s_waitcnt vmcnt(0)
      // round trips in series.
      const Tri t = tri_setup(z, y, x);
      int64_t zo[2], yo[2], xo[2];
      float wzv[2], wyv[2], wxv[2];
      bool zkv[2], ykv[2], xkv[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int zz = t.z0 + d, yy = t.y0 + d, xx = t.x0 + d;
        const bool zk = zz >= 0 && zz < D, yk = yy >= 0 && yy < H, xk = xx >= 0 && xx < W;
        zkv[d] = zk; ykv[d] = yk; xkv[d] = xk;
        zo[d] = (int64_t)(zk ? zz : 0) * H * W * C;
        yo[d] = (int64_t)(yk ? yy : 0) * W * C;
        xo[d] = (int64_t)(xk ? xx : 0) * C;
        wzv[d] = d ? t.fz : 1.f - t.fz;
        wyv[d] = d ? t.fy : 1.f - t.fy;
        wxv[d] = d ? t.fx : 1.f - t.fx;
      }
      float s[8][CPT];
#pragma unroll
      for (int q = 0; q < 8; ++q) ldsrc<CPT, S16>(src, sbe + zo[q >> 2] + yo[(q >> 1) & 1] + xo[q & 1], s[q]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float wgt = wzv[q >> 2] * wyv[(q >> 1) & 1] * wxv[q & 1];
        const bool ok = zkv[q >> 2] && ykv[(q >> 1) & 1] && xkv[q & 1];
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[c] = fmaf(wgt, ok ? s[q][c] : 0.f, acc[c]);
      }
    }
    if (add_flow) {                            // C == 3, CPT == 3, G == 1
      if constexpr (CPT == 3) { acc[0] += f0; acc[1] += f1; acc[2] += f2; }
    }
    if constexpr (O16 && CPT == 4) {
      const __bf16 b0 = (__bf16)acc[0], b1 = (__bf16)acc[1], b2 = (__bf16)acc[2], b3 = (__bf16)acc[3];
      const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
      const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + n * C + g * CPT) = make_uint2(lo, hi);
    } else {
      stv<CPT>(out + n * C + g * CPT, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------ warp bwd
// d_src is a scatter-add of the 8 corner weights (float atomics, as ATen's grid_sampler_3d_backward does).
// gfx950's L2 retires ~1 dword atomic per clock per channel (~270 G/s measured), so what matters is how many
// distinct 64 B lines one wave instruction touches: threads map to (voxel, channel) with the channel fastest,
// so the lanes of one atomic instruction hit whole contiguous channel vectors (measured on C=8, 160x192x160:
// 1.4-1.9 ms for smooth AND rough flows vs 5.4-7.5 ms with 4 channels per thread; an LDS-privatised variant was
// 1.7 ms on smooth but 14 ms on rough flows and was dropped).  d_flow: per-voxel gather; the G (= C rounded up
// to a power of two) channel lanes of a voxel are adjacent and combined with xor-shuffles.
// Threads <-> (voxel, channel); a thread walks a run of ZRUN voxels along z.  Two merges cut the L2 atomics (the
// bound of this kernel, ~270 G dword-atomics/s) without any LDS:
//  * x: the next voxel along x sits G lanes up.  When its footprint is this one shifted by +1 in x (the common case
//    for a smooth flow) its dx=0 corners are this voxel's dx=1 corners: hand those four values over with a shuffle
//    and let the neighbour issue ONE atomic for both.  Both sides evaluate the same predicate from shuffled
//    footprints, so nothing is lost or counted twice.
//  * z: the dz=1 half of a voxel's footprint is kept in registers; if the next voxel of the run lands exactly one
//    source plane higher it absorbs them into its dz=0 half, otherwise they are flushed.
// Smooth flow: ~2 atomics per (voxel, channel) instead of 8.
constexpr int ZRUN = 8;
// DETERMINISTIC form of the scatter (round 5, opt-in: modet_warp_bwd_det): float atomics make d_src depend on the order the
// hardware retires them (1e-6 run to run -- as ATen's grid_sampler_3d_backward, reference ModeT/models.py:67).  Here every
// contribution is converted to a 64-bit FIXED-POINT integer (scale = a power of two chosen from max |d_out| so that 2^23
// contributions of the largest size fit) and added with an integer atomic: integer addition is associative, so the sums --
// and with them the whole train step, whose only atomics these are -- are bit-identical from run to run.  What a thread adds
// (its merged carries) is itself a fixed sequence of float operations.  Resolution 2^-40 of max |d_out|: finer than fp32.
// Costs the max pre-pass, an 8-byte accumulator per element and the decode pass (tests, debugging, exact reproduction).
__device__ __forceinline__ double det_scale(const unsigned* header) {
  const float m = __uint_as_float(header[0]);                       // max |d_out| (bit pattern: atomicMax on the uint is exact)
  int e;
  frexpf(m > 0.f ? m : 1.f, &e);                                    // m < 2^e
  return ldexp(1.0, 40 - e);
}
template <bool DET>
__device__ __forceinline__ void scat_add(float* p, float v, const float* base, long long* acc, double scale) {
  if constexpr (DET) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + (p - base)), (unsigned long long)__double2ll_rn((double)v * scale));
  } else {
    atomicAdd(p, v);
  }
}
__global__ __launch_bounds__(BLK) void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ header) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(header, __float_as_uint(m));
}
__global__ __launch_bounds__(BLK) void det_decode_kernel(const long long* __restrict__ acc, float* __restrict__ out, int64_t n,
                                                         const unsigned* __restrict__ header) {
  const double inv = 1.0 / det_scale(header);
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) out[i] = (float)((double)acc[i] * inv);
}

template <bool S16, bool DET = false>
__global__ __launch_bounds__(BLK) void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                       const float* __restrict__ dout, float* __restrict__ dsrc,
                                                       float* __restrict__ dflow, const float* __restrict__ dfadd, int D, int H,
                                                       int W, int C, int G, int64_t total, int add_flow,
                                                       long long* __restrict__ dacc = nullptr, const unsigned* __restrict__ dhdr = nullptr) {
  // total = B * ceil(D/ZRUN) * H * W * G items (one per run)
  const double dscale = DET ? det_scale(dhdr) : 0.0;
  const int64_t V = (int64_t)D * H * W;
  const int nrun = (D + ZRUN - 1) / ZRUN;
  const int64_t HW = (int64_t)H * W;
  const int64_t total_pad = cdiv64(total, BLK) * BLK;       // keep whole waves alive for the shuffles
  const int lane = threadIdx.x & 63;
  for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total_pad; idx += (int64_t)gridDim.x * BLK) {
    const bool inr = idx < total;
    const int64_t id = inr ? idx : total - 1;
    const int c = (int)(id % G);
    const bool livec = inr && c < C;
    int64_t r = id / G;
    const int xi = (int)(r % W); r /= W;
    const int yi = (int)(r % H); r /= H;
    const int zr = (int)(r % nrun);
    const int64_t b = r / nrun;
    const int cc = livec ? c : 0;
    const int64_t sbe = b * V * C + cc;            // element offset inside src (fp32 or, S16, bf16)
    float* db = dsrc ? dsrc + b * V * C + cc : nullptr;
    // x-neighbour bookkeeping that does not depend on z
    const int up = lane + G, dn = lane - G;
    const int nxi = __shfl(xi, up, 64), pxi = __shfl(xi, dn, 64);
    const bool up_ok = (up < 64) && (idx + G < total) && nxi == xi + 1;
    const bool dn_ok = (dn >= 0) && inr && pxi == xi - 1;
    float pend[4] = {0.f, 0.f, 0.f, 0.f};                    // dz=1 half of the previous voxel, index dy*2 + dx
    int pz = 0, py = 0, px = 0;
    int64_t poff = 0;                                        // element offset of corner (pz, py, px) in a channel's volume
    bool have = false;
    // corner addressing: ONE 64-bit multiply per voxel (the element offset of its (z0, y0, x0) corner), the other seven
    // corners by adding wave-uniform strides -- written per corner as (((z * H + y) * W + x) * C the address arithmetic
    // (quarter-rate v_mul_lo_u32 / v_mad_u64_u32, ~150 per voxel) was most of this kernel's instructions
    const int64_t sXc = C, sYc = (int64_t)W * C, sZc = HW * C;
    // Every load is unconditional (clamped, always valid addresses; the predicate goes into the VALUE): a load under
    // `if (ok)` costs its own s_waitcnt vmcnt(0) -- the first version waited for ten memory round trips in series per
    // voxel (flow, d_out, then the eight source corners one by one).  flow and d_out of the NEXT voxel of the run are
    // fetched while the current one is processed.
    const int64_t nrow = b * V + (int64_t)yi * W + xi;
    float pf0, pf1, pf2, pgo;
    {
      const int z0i = zr * ZRUN < D ? zr * ZRUN : D - 1;
      const int64_t n0 = nrow + (int64_t)z0i * HW;
      pf0 = flow[n0 * 3]; pf1 = flow[n0 * 3 + 1]; pf2 = flow[n0 * 3 + 2];
      pgo = dout[n0 * C + cc];
    }
    for (int k = 0; k < ZRUN; ++k) {
      const int zi = zr * ZRUN + k;
      const bool zin = zi < D;                               // uniform over the wave except at the run tail
      const bool live = livec && zin;
      const int64_t n = nrow + (int64_t)(zin ? zi : D - 1) * HW;
      const float f0 = pf0, f1 = pf1, f2 = pf2;
      const float go = live ? pgo : 0.f;
      {
        const int zn = (zi + 1 < D && k + 1 < ZRUN) ? zi + 1 : (zin ? zi : D - 1);
        const int64_t nn = nrow + (int64_t)zn * HW;
        pf0 = flow[nn * 3]; pf1 = flow[nn * 3 + 1]; pf2 = flow[nn * 3 + 2];
        pgo = dout[nn * C + cc];
      }
      const Tri t = tri_setup((float)zi + f0, (float)yi + f1, (float)xi + f2);
      // contributions of this (voxel, channel) to the 8 corners, index = dz*4 + dy*2 + dx
      float cv[8];
      float gz = 0.f, gy = 0.f, gx = 0.f;
      const int64_t off0 = (((int64_t)t.z0 * H + t.y0) * W + t.x0) * C;
      const bool zok[2] = {t.z0 >= 0 && t.z0 < D, t.z0 + 1 >= 0 && t.z0 + 1 < D};
      const bool yok[2] = {t.y0 >= 0 && t.y0 < H, t.y0 + 1 >= 0 && t.y0 + 1 < H};
      const bool xok[2] = {t.x0 >= 0 && t.x0 < W, t.x0 + 1 >= 0 && t.x0 + 1 < W};
      float sv[8];
      if (dflow) {                                           // uniform
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool ok = live && zok[q >> 2] && yok[(q >> 1) & 1] && xok[q & 1];
          const int64_t off = off0 + ((q >> 2) ? sZc : 0) + (((q >> 1) & 1) ? sYc : 0) + ((q & 1) ? sXc : 0);
          if constexpr (S16) sv[q] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(src)[sbe + (ok ? off : 0)] << 16);
          else sv[q] = src[sbe + (ok ? off : 0)];
        }
      }
#pragma unroll
      for (int dz = 0; dz < 2; ++dz) {
        const float wz = dz ? t.fz : 1.f - t.fz;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const float wy = dy ? t.fy : 1.f - t.fy;
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const float wx = dx ? t.fx : 1.f - t.fx;
            const bool ok = live && zok[dz] && yok[dy] && xok[dx];
            cv[dz * 4 + dy * 2 + dx] = ok ? wz * wy * wx * go : 0.f;
            if (dflow) {
              const float dot = ok ? sv[dz * 4 + dy * 2 + dx] * go : 0.f;     // an exact zero for the corners outside
              gz += (dz ? 1.f : -1.f) * wy * wx * dot;
              gy += (dy ? 1.f : -1.f) * wz * wx * dot;
              gx += (dx ? 1.f : -1.f) * wz * wy * dot;
            }
          }
        }
      }
      if (db) {
        // ---- x merge
        const int nz0 = __shfl(t.z0, up, 64), ny0 = __shfl(t.y0, up, 64), nx0 = __shfl(t.x0, up, 64);
        const bool give = up_ok && zin && nz0 == t.z0 && ny0 == t.y0 && nx0 == t.x0 + 1;
        const int pz0 = __shfl(t.z0, dn, 64), py0 = __shfl(t.y0, dn, 64), px0 = __shfl(t.x0, dn, 64);
        const bool take = dn_ok && zin && pz0 == t.z0 && py0 == t.y0 && px0 == t.x0 - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                        // q = dz*2 + dy
          const float from_prev = __shfl(cv[q * 2 + 1], dn, 64);
          if (take) cv[q * 2] += from_prev;
          if (give) cv[q * 2 + 1] = 0.f;
        }
        // ---- z merge
        if (have) {
          if (zin && t.z0 == pz && t.y0 == py && t.x0 == px) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cv[q] += pend[q];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (pend[q] != 0.f) scat_add<DET>(db + poff + ((q >> 1) ? sYc : 0) + ((q & 1) ? sXc : 0), pend[q], dsrc, dacc, dscale);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                        // the dz=0 half is final now
          if (cv[q] != 0.f) scat_add<DET>(db + off0 + ((q >> 1) ? sYc : 0) + ((q & 1) ? sXc : 0), cv[q], dsrc, dacc, dscale);
          pend[q] = cv[4 + q];
        }
        pz = t.z0 + 1; py = t.y0; px = t.x0; poff = off0 + sZc;
        have = zin;
      }
      if (dflow) {
        if (add_flow && live) {             // C == 3: d(out_c)/d(flow_c) has the identity term
          gz += c == 0 ? go : 0.f; gy += c == 1 ? go : 0.f; gx += c == 2 ? go : 0.f;
        }
        for (int o = 1; o < G; o <<= 1) {
          gz += __shfl_xor(gz, o, 64);
          gy += __shfl_xor(gy, o, 64);
          gx += __shfl_xor(gx, o, 64);
        }
        if (inr && zin && c == 0) {
          float* dfp = dflow + n * 3;
          if (dfadd) { const float* ap = dfadd + n * 3; gz += ap[0]; gy += ap[1]; gx += ap[2]; }     // (uniform)
          dfp[0] = gz; dfp[1] = gy; dfp[2] = gx;
        }
      }
    }
    if (db && have) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (pend[q] != 0.f) scat_add<DET>(db + poff + ((q >> 1) ? sYc : 0) + ((q & 1) ? sXc : 0), pend[q], dsrc, dacc, dscale);
    }
  }
}

// Second form of the same backward with a THIRD merge (y), for tensors below 2^31 elements (32-bit offsets): a thread owns
// one (x, channel lane) and walks a PATCH of ZR planes x YR rows, z outer (run-time loop), y inner (unrolled).  Contributions
// travel in two kinds of carries, each tagged with the LINEAR element offset of the cell it belongs to:
//   * inner (y) carry: the four (dz, dx) values of the voxel's dy = 1 level, merged into the next voxel of the run when that
//     one's base cell is exactly this offset (otherwise flushed);
//   * outer (z) carry: per position of the y run, the two (dx) values of the (dz = 1, dy = 0) level (already holding what the
//     y merge put there), consumed by the voxel at the same position of the NEXT plane when its base cell matches.
// Equality of linear offsets is sufficient: a carried value is non-zero only for a cell inside the volume, and a value is only
// ever added to something that is then written at that same linear offset.  On a smooth flow every run of YR voxels issues
// YR + 1 atomics for its dz = 0 level (+ one outer flush per patch): ~1.4 per (voxel, channel) against 2.25 with the x / z
// merges of warp_bwd_kernel.  Same arithmetic per contribution; the order of the float adds differs (as between any two runs
// of the atomic form).
constexpr int WB_NONE = -2147483647 - 1;
constexpr unsigned WB_OOB = 0x80000000u;               // buffer offset past any tensor here (< 2 GiB, checked on the host)
#ifndef WB2_ZR
#define WB2_ZR 8
#endif
#ifndef WB2_YR
#define WB2_YR 4
#endif
#ifdef MODET_TUNING
__device__ unsigned long long* g_wb_dbg = nullptr;
#endif
using WbRsrc = __amdgpu_buffer_rsrc_t;
using wb_u32x3 = unsigned __attribute__((ext_vector_type(3)));
__device__ __forceinline__ WbRsrc wb_rsrc(const void* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// Instruction diet of the first patch kernel (~750 instructions per voxel, 0.86 ms at level 1 whatever the patch shape: issue
// bound, not atomic bound): every tensor through a buffer descriptor with a 32-bit byte offset (no 64-bit address pairs; an
// offset past the tensor reads 0 / drops the atomic, so "this lane has nothing to add" is an offset select, not a branch),
// 24-bit integer multiplies (full rate; v_mul_lo_u32 is quarter rate), per-axis validity as -1 / 0 masks folded into the
// weights, flow as ONE 12-byte load, the x neighbour test on ONE shuffled value (the cell's byte offset: equal offsets = same
// cell, see above), d_flow sums as (upper - lower) differences.
template <int ZR, int YR, bool S16, bool DET = false>
__global__ __launch_bounds__(BLK) void warp_bwd2_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                        const float* __restrict__ dout, float* __restrict__ dsrc,
                                                        float* __restrict__ dflow, const float* __restrict__ dfadd, int B, int D,
                                                        int H, int W, int C, int G, unsigned total, int add_flow,
                                                        long long* __restrict__ dacc = nullptr, const unsigned* __restrict__ dhdr = nullptr) {
  // total = B * ceil(D/ZR) * ceil(H/YR) * W * G items (one per patch column)
  const double dscale = DET ? det_scale(dhdr) : 0.0;
  const int V = D * H * W;
  const unsigned nzr = (unsigned)((D + ZR - 1) / ZR), nyr = (unsigned)((H + YR - 1) / YR);
  const unsigned total_pad = (total + BLK - 1) / BLK * BLK;       // keep whole waves alive for the shuffles
  const int lane = threadIdx.x & 63;
  const int sX = C * 4, sY = W * C * 4, sZ = H * W * C * 4;       // byte strides of a channel's volume
  const unsigned tbytes = (unsigned)B * (unsigned)V * (unsigned)C * 4u, fbytes = (unsigned)B * (unsigned)V * 12u;
  const WbRsrc r_src = wb_rsrc(src, S16 ? tbytes / 2 : tbytes), r_flow = wb_rsrc(flow, fbytes), r_do = wb_rsrc(dout, tbytes);
  const WbRsrc r_ds = wb_rsrc(dsrc, dsrc ? tbytes : 0u), r_df = wb_rsrc(dflow, dflow ? fbytes : 0u);
  const WbRsrc r_da = wb_rsrc(dfadd ? dfadd : flow, dfadd ? fbytes : 0u);    // a second gradient of the same flow, added on the way out (0 bytes: reads 0)
  const bool want_src = dsrc != nullptr, want_flow = dflow != nullptr;
  for (unsigned idx = blockIdx.x * BLK + threadIdx.x; idx < total_pad; idx += gridDim.x * BLK) {
    const bool inr = idx < total;
    const unsigned id = inr ? idx : total - 1;
    const int c = (int)(id % (unsigned)G);
    const bool livec = inr && c < C;
    unsigned r = id / (unsigned)G;
    const int xi = (int)(r % (unsigned)W); r /= (unsigned)W;
    const int yr = (int)(r % nyr); r /= nyr;
    const int zr = (int)(r % nzr);
    const int b = (int)(r / nzr);
    const int cc = livec ? c : 0;
    const int cb = (b * V * C + cc) * 4;               // byte offset of (sample, channel) inside src / d_src / d_out
    const int up = lane + G, dn = lane - G;
    const int nxi = __shfl(xi, up, 64), pxi = __shfl(xi, dn, 64);
    const bool up_ok = (up < 64) && (idx + G < total) && nxi == xi + 1;
    const bool dn_ok = (dn >= 0) && inr && pxi == xi - 1;
    float OC[YR + 1][2];                               // outer carry: (dz = 1, dy = 0) level per position of the y run, [dx]
    int OCoff[YR + 1];
#pragma unroll
    for (int k = 0; k <= YR; ++k) { OC[k][0] = 0.f; OC[k][1] = 0.f; OCoff[k] = WB_NONE; }
    auto atom = [&](int off, float v) {                // branch-free: nothing to add = an out-of-range offset
#ifdef MODET_TUNING
      if (g_wb_dbg) {                                  // census: atomic instructions issued, lanes that add something
        const unsigned long long m = __builtin_amdgcn_ballot_w64(v != 0.f);
        if (lane == 0) { atomicAdd(g_wb_dbg, 1ull); atomicAdd(g_wb_dbg + 1, (unsigned long long)__builtin_popcountll(m)); }
      }
#endif
      if constexpr (DET) {                             // (see scat_add: 64-bit fixed point, order-independent)
        if (v != 0.f) atomicAdd(reinterpret_cast<unsigned long long*>(dacc) + ((unsigned)off >> 2),
                                (unsigned long long)__double2ll_rn((double)v * dscale));
      } else {
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r_ds, v != 0.f ? (unsigned)off : WB_OOB, 0, 0);
      }
    };
    auto flush2 = [&](int off, float v0, float v1) {   // (for the rarely taken paths: skipped when no lane has anything)
      if (v0 != 0.f) atom(off, v0);
      if (v1 != 0.f) atom(off + sX, v1);
    };
    const int y0p = yr * YR;
#pragma unroll 1
    for (int j = 0; j < ZR; ++j) {
      const int zi = zr * ZR + j;
      const bool zin = zi < D;
      const int zc = zin ? zi : D - 1;
      float IC[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // inner carry: [dz][dx] of the previous voxel's dy = 1 level
      int ICoff = WB_NONE;
      float ON[YR + 1][2];
      int ONoff[YR + 1];
      const int nplane = b * V + __mul24(zc, H * W) + xi;
      // all loads of the run up front (flow, d_out of its YR voxels): independent of each other
      wb_u32x3 fv[YR];
      float gov[YR];
#pragma unroll
      for (int k = 0; k < YR; ++k) {
        const int yk = y0p + k < H ? y0p + k : H - 1;
        const int n = nplane + __mul24(yk, W);
        fv[k] = __builtin_amdgcn_raw_buffer_load_b96(r_flow, (unsigned)n * 12u, 0, 0);
        gov[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_do, (unsigned)n * (unsigned)sX + (unsigned)cc * 4u, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < YR; ++k) {
        const int yi = y0p + k;
        const bool yin = yi < H;
        const int yc = yin ? yi : H - 1;
        const bool live = livec && zin && yin;
        const int n = nplane + __mul24(yc, W);
        const float go = live ? gov[k] : 0.f;
        // ---- footprint: floor, fraction, base cell clamped to [-2, dim] (anything outside [-1, dim - 1] has no valid corner)
        const float z = (float)zc + __uint_as_float(fv[k][0]), y = (float)yc + __uint_as_float(fv[k][1]), x = (float)xi + __uint_as_float(fv[k][2]);
        const float zf = floorf(z), yf = floorf(y), xf = floorf(x);
        const float fz = z - zf, fy = y - yf, fx = x - xf;
        const int z0 = (int)fminf(fmaxf(zf, -2.f), (float)D), y0 = (int)fminf(fmaxf(yf, -2.f), (float)H), x0 = (int)fminf(fmaxf(xf, -2.f), (float)W);
        const int off0 = __mul24(__mul24(__mul24(z0, H) + y0, W) + x0, sX) + cb;      // byte offset of the (z0, y0, x0) cell
        // per-axis validity as all-ones / zero masks, folded into the weights
        const int mz0 = (unsigned)z0 < (unsigned)D ? -1 : 0, mz1 = (unsigned)(z0 + 1) < (unsigned)D ? -1 : 0;
        const int my0 = (unsigned)y0 < (unsigned)H ? -1 : 0, my1 = (unsigned)(y0 + 1) < (unsigned)H ? -1 : 0;
        const int mx0 = (unsigned)x0 < (unsigned)W ? -1 : 0, mx1 = (unsigned)(x0 + 1) < (unsigned)W ? -1 : 0;
        auto msk = [](float v, int m) { return __uint_as_float(__float_as_uint(v) & (unsigned)m); };
        const float wz[2] = {msk(1.f - fz, mz0), msk(fz, mz1)};
        const float wy[2] = {msk(1.f - fy, my0), msk(fy, my1)};
        const float wx[2] = {msk(1.f - fx, mx0), msk(fx, mx1)};
        const float pyx[4] = {wy[0] * wx[0], wy[0] * wx[1], wy[1] * wx[0], wy[1] * wx[1]};     // [dy*2 + dx]
        float cv[8];                                   // index dz*4 + dy*2 + dx
        {
          const float g0 = wz[0] * go, g1 = wz[1] * go;
#pragma unroll
          for (int q = 0; q < 4; ++q) { cv[q] = g0 * pyx[q]; cv[4 + q] = g1 * pyx[q]; }
        }
        float gz = 0.f, gy = 0.f, gx = 0.f;
        if (want_flow) {                               // uniform
          float dot[8];
          const int mzy[4] = {mz0 & my0, mz0 & my1, mz1 & my0, mz1 & my1};
#pragma unroll
          for (int q = 0; q < 8; ++q) {                // an offset outside the tensor reads 0; an aliased one is masked
            const int off = off0 + ((q >> 2) ? sZ : 0) + (((q >> 1) & 1) ? sY : 0) + ((q & 1) ? sX : 0);
            float sv;                                  // (offsets are bytes of the fp32 layout: a bf16 src sits at half of them)
            if constexpr (S16) sv = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r_src, (unsigned)off >> 1, 0, 0) << 16);
            else sv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_src, (unsigned)off, 0, 0));
            dot[q] = msk(sv, mzy[q >> 1] & ((q & 1) ? mx1 : mx0)) * go;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) gz = fmaf(pyx[q], dot[4 + q] - dot[q], gz);
          const float wzx[4] = {wz[0] * wx[0], wz[0] * wx[1], wz[1] * wx[0], wz[1] * wx[1]};     // [dz*2 + dx]
          const float wzy[4] = {wz[0] * wy[0], wz[0] * wy[1], wz[1] * wy[0], wz[1] * wy[1]};     // [dz*2 + dy]
#pragma unroll
          for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              gy = fmaf(wzx[dz * 2 + d], dot[dz * 4 + 2 + d] - dot[dz * 4 + d], gy);          // d = dx
              gx = fmaf(wzy[dz * 2 + d], dot[dz * 4 + d * 2 + 1] - dot[dz * 4 + d * 2], gx);  // d = dy
            }
        }
        if (want_src) {
          // ---- x merge: the neighbour one voxel up in x takes this voxel's dx = 1 corners when its base cell is this one + 1
          const int nup = __shfl(off0, up, 64), ndn = __shfl(off0, dn, 64);
          const bool give = up_ok && zin && yin && nup == off0 + sX;
          const bool take = dn_ok && zin && yin && ndn + sX == off0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {                // q = dz*2 + dy
            const float from_prev = __shfl(cv[q * 2 + 1], dn, 64);
            cv[q * 2] += take ? from_prev : 0.f;
            cv[q * 2 + 1] = give ? 0.f : cv[q * 2 + 1];
          }
          // ---- z merge: the carry of this run position from the previous plane, (dz = 0, dy = 0) of this voxel
          if (OCoff[k] == off0) { cv[0] += OC[k][0]; cv[1] += OC[k][1]; }
          else flush2(OCoff[k], OC[k][0], OC[k][1]);
          if (k == YR - 1) {                           // the run's last carry meets the last voxel's (dz = 0, dy = 1)
            if (OCoff[YR] == off0 + sY) { cv[2] += OC[YR][0]; cv[3] += OC[YR][1]; }
            else flush2(OCoff[YR], OC[YR][0], OC[YR][1]);
          }
          // ---- y merge: the previous voxel's dy = 1 level
          if (ICoff == off0) { cv[0] += IC[0][0]; cv[1] += IC[0][1]; cv[4] += IC[1][0]; cv[5] += IC[1][1]; }
          else { flush2(ICoff, IC[0][0], IC[0][1]); flush2(ICoff + sZ, IC[1][0], IC[1][1]); }
          // ---- (dz = 0, dy = 0) is final; (dz = 1, dy = 0) waits for the next plane; dy = 1 for the next voxel of the run
          atom(off0, cv[0]);
          if (__builtin_amdgcn_ballot_w64(cv[1] != 0.f)) atom(off0 + sX, cv[1]);     // (only where the x neighbour did not take it)
          ON[k][0] = cv[4]; ON[k][1] = cv[5]; ONoff[k] = off0 + sZ;
          IC[0][0] = cv[2]; IC[0][1] = cv[3]; IC[1][0] = cv[6]; IC[1][1] = cv[7];
          ICoff = off0 + sY;
        }
        if (want_flow) {
          if (add_flow && live) {                      // C == 3: d(out_c)/d(flow_c) has the identity term
            gz += c == 0 ? go : 0.f; gy += c == 1 ? go : 0.f; gx += c == 2 ? go : 0.f;
          }
          for (int o = 1; o < G; o <<= 1) {
            gz += __shfl_xor(gz, o, 64);
            gy += __shfl_xor(gy, o, 64);
            gx += __shfl_xor(gx, o, 64);
          }
          const unsigned dfo = (inr && zin && yin && c == 0) ? (unsigned)n * 12u : WB_OOB;
          const wb_u32x3 ad = __builtin_amdgcn_raw_buffer_load_b96(r_da, dfo, 0, 0);
          const wb_u32x3 g3 = {__float_as_uint(gz + __uint_as_float(ad[0])), __float_as_uint(gy + __uint_as_float(ad[1])),
                               __float_as_uint(gx + __uint_as_float(ad[2]))};
          __builtin_amdgcn_raw_buffer_store_b96(g3, r_df, dfo, 0, 0);
        }
      }
      if (want_src) {                                  // run end: the last voxel's dy = 1 level
        atom(ICoff, IC[0][0]);
        if (__builtin_amdgcn_ballot_w64(IC[0][1] != 0.f)) atom(ICoff + sX, IC[0][1]);
        ON[YR][0] = IC[1][0]; ON[YR][1] = IC[1][1]; ONoff[YR] = ICoff == WB_NONE ? WB_NONE : ICoff + sZ;
#pragma unroll
        for (int k = 0; k <= YR; ++k) { OC[k][0] = ON[k][0]; OC[k][1] = ON[k][1]; OCoff[k] = ONoff[k]; }
      }
    }
    if (want_src) {
#pragma unroll
      for (int k = 0; k <= YR; ++k) { atom(OCoff[k], OC[k][0]); atom(OCoff[k] + sX, OC[k][1]); }
    }
  }
}

// Bounded-flow backward (|flow| <= 1, C == 3: the flow compositions with an attention output).  A sample point
// p + flow[p] lies within one voxel of p, so source voxel s only receives from the 27 voxels p = s + d, d in {-1,0,1}^3,
// with weight prod_a max(0, 1 - |p_a + flow_a[p] - s_a|) (the trilinear hat: (1-f) at floor, f at floor+1).
// One thread per voxel does both jobs: gathers d_src[s] (no atomics, deterministic) and, as p, its own d_flow.
// Runs on 4x4x16 voxel tiles: flow and d_out of the tile + 1-voxel halo are staged in LDS once (6 floats per voxel;
// d_out = 0 outside the volume, so the 27-neighbour loop needs no bounds tests).
constexpr int G3Z = 4, G3Y = 4, G3X = 16, G3HZ = G3Z + 2, G3HY = G3Y + 2, G3HX = G3X + 2, G3HV = G3HZ * G3HY * G3HX;
__global__ __launch_bounds__(BLK) void warp_bwd_gather3_tiled_kernel(const float* __restrict__ src,
                                                                     const float* __restrict__ flow,
                                                                     const float* __restrict__ dout,
                                                                     float* __restrict__ dsrc, float* __restrict__ dflow,
                                                                     int D, int H, int W, int tiles_x, int tiles_y,
                                                                     int add_flow) {
  __shared__ __attribute__((aligned(8))) float fg[G3HV * 6];
  const int64_t V = (int64_t)D * H * W;
  const int b = blockIdx.y;
  int t = blockIdx.x;
  const int x0 = (t % tiles_x) * G3X; t /= tiles_x;
  const int y0 = (t % tiles_y) * G3Y;
  const int z0 = (t / tiles_y) * G3Z;
  // staging, branch-free: all loads of a thread are issued before the first LDS write (clamped addresses, the halo
  // outside the volume selected to zero afterwards); with a load per `if (inside)` every item was its own round trip
  constexpr int G3N = (G3HV * 3 + BLK - 1) / BLK;
  float2 stg[G3N];
#pragma unroll
  for (int j = 0; j < G3N; ++j) {
    const int i = min((int)threadIdx.x + j * BLK, G3HV * 3 - 1);
    const int v = i / 3, part = i - v * 3;               // part 0: flow[0..1]; 1: flow[2], d_out[0]; 2: d_out[1..2]
    const int hx = v % G3HX, r = v / G3HX;
    const int hy = r % G3HY, hz = r / G3HY;
    const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
    const bool in = z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
    const int64_t n = (int64_t)b * V + (in ? ((int64_t)z * H + y) * W + x : 0);
    const float* pa = part == 0 ? flow + n * 3 : (part == 1 ? flow + n * 3 + 2 : dout + n * 3 + 1);
    const float* pb = part == 0 ? flow + n * 3 + 1 : (part == 1 ? dout + n * 3 : dout + n * 3 + 2);
    const float va = *pa, vb = *pb;
    stg[j] = in ? make_float2(va, vb) : make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < G3N; ++j) {
    const int i = threadIdx.x + j * BLK;
    if (i < G3HV * 3) *reinterpret_cast<float2*>(fg + i * 2) = stg[j];       // fg + v * 6 + part * 2 with i = v * 3 + part
  }
  __syncthreads();
  const int tx = threadIdx.x % G3X, ty = (threadIdx.x / G3X) % G3Y, tz = threadIdx.x / (G3X * G3Y);
  const int zi = z0 + tz, yi = y0 + ty, xi = x0 + tx;
  if (zi >= D || yi >= H || xi >= W) return;
  const int64_t n = (int64_t)b * V + ((int64_t)zi * H + yi) * W + xi;
  const int lc = ((tz + 1) * G3HY + ty + 1) * G3HX + tx + 1;
  if (dsrc) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const float* p = fg + (lc + (dz * G3HY + dy) * G3HX + dx) * 6;
          const float2 q0 = *reinterpret_cast<const float2*>(p), q1 = *reinterpret_cast<const float2*>(p + 2),
                       q2 = *reinterpret_cast<const float2*>(p + 4);
          const float wz = 1.f - fabsf((float)dz + q0.x);
          const float wy = 1.f - fabsf((float)dy + q0.y);
          const float wx = 1.f - fabsf((float)dx + q1.x);
          if (wz > 0.f && wy > 0.f && wx > 0.f) {
            const float wgt = wz * wy * wx;
            a0 = fmaf(wgt, q1.y, a0); a1 = fmaf(wgt, q2.x, a1); a2 = fmaf(wgt, q2.y, a2);
          }
        }
    float* dp = dsrc + n * 3;
    dp[0] = a0; dp[1] = a1; dp[2] = a2;
  }
  if (dflow) {
    const float* pc = fg + lc * 6;
    const Tri tr = tri_setup((float)zi + pc[0], (float)yi + pc[1], (float)xi + pc[2]);
    const float g0 = pc[3], g1 = pc[4], g2 = pc[5];
    const float* sb = src + (int64_t)b * V * 3;
    float gz = 0.f, gy = 0.f, gx = 0.f;
    // eight corner gathers issued back to back (clamped address + zero weight outside; see warp_fwd_kernel)
    float sp[8][3];
    bool okc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int zz = tr.z0 + (q >> 2), yy = tr.y0 + ((q >> 1) & 1), xx = tr.x0 + (q & 1);
      okc[q] = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float* pq = sb + (okc[q] ? (((int64_t)zz * H + yy) * W + xx) * 3 : 0);
      sp[q][0] = pq[0]; sp[q][1] = pq[1]; sp[q][2] = pq[2];
    }
#pragma unroll
    for (int dz = 0; dz < 2; ++dz) {
      const float wz = dz ? tr.fz : 1.f - tr.fz;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const float wy = dy ? tr.fy : 1.f - tr.fy;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const float wx = dx ? tr.fx : 1.f - tr.fx;
          const int q = dz * 4 + dy * 2 + dx;
          const float dot = okc[q] ? sp[q][0] * g0 + sp[q][1] * g1 + sp[q][2] * g2 : 0.f;
          gz += (dz ? 1.f : -1.f) * wy * wx * dot;
          gy += (dy ? 1.f : -1.f) * wz * wx * dot;
          gx += (dx ? 1.f : -1.f) * wz * wy * dot;
        }
      }
    }
    if (add_flow) { gz += g0; gy += g1; gx += g2; }
    float* dfp = dflow + n * 3;
    dfp[0] = gz; dfp[1] = gy; dfp[2] = gx;
  }
}

// ------------------------------------------------------------------------------------------------ upsample x2
struct Lin { int i0, i1; float l0, l1; };
// ATen compute_source_index_and_lambda, align_corners=True (UpSample.h): src = ratio*dst
__device__ __forceinline__ Lin lin_src(int o, float ratio, int n_in) {
  Lin r;
  const float real = ratio * (float)o;
  r.i0 = min((int)real, n_in - 1);
  r.l1 = fminf(fmaxf(real - (float)r.i0, 0.f), 1.f);
  r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
  r.l0 = 1.f - r.l1;
  return r;
}
__host__ __device__ __forceinline__ float up_ratio(int n_in) {
  return n_in > 1 ? (float)(n_in - 1) / (float)(2 * n_in - 1) : 0.f;
}

template <int CPT>
__global__ __launch_bounds__(BLK) void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int d,
                                                            int h, int w, int C, int G, int64_t total, float scale) {
  const int D = 2 * d, H = 2 * h, W = 2 * w;
  const float rz = up_ratio(d), ry = up_ratio(h), rx = up_ratio(w);
  const int64_t Vo = (int64_t)D * H * W, Vi = (int64_t)d * h * w;
  const bool small = total < (1ll << 31) && Vo < (1ll << 31);
  for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
    int g, xo, yo, zo;
    int64_t n, b;
    if (small) {               // 32-bit index arithmetic: 64-bit div/mod are ~100-instruction software routines
      const unsigned u = (unsigned)idx, un = u / (unsigned)G;
      g = (int)(u - un * (unsigned)G);
      const unsigned ub = un / (unsigned)Vo, uv = un - ub * (unsigned)Vo;
      const unsigned t2 = uv / (unsigned)W;
      xo = (int)(uv - t2 * (unsigned)W);
      zo = (int)(t2 / (unsigned)H);
      yo = (int)(t2 - (unsigned)zo * (unsigned)H);
      n = un; b = ub;
    } else {
      g = (int)(idx % G);
      n = idx / G;
      b = n / Vo;
      const int64_t v = n - b * Vo;
      xo = (int)(v % W);
      const int64_t t2 = v / W;
      yo = (int)(t2 % H); zo = (int)(t2 / H);
    }
    const Lin lz = lin_src(zo, rz, d), ly = lin_src(yo, ry, h), lx = lin_src(xo, rx, w);
    const float* xb = x + b * Vi * C + g * CPT;
    float acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int zz = a ? lz.i1 : lz.i0, yy = bb ? ly.i1 : ly.i0, xx = cc ? lx.i1 : lx.i0;
          const float wgt = (a ? lz.l1 : lz.l0) * (bb ? ly.l1 : ly.l0) * (cc ? lx.l1 : lx.l0);
          float s[CPT];
          ldv<CPT>(xb + (((int64_t)zz * h + yy) * w + xx) * C, s);
#pragma unroll
          for (int c = 0; c < CPT; ++c) acc[c] = fmaf(wgt, s[c], acc[c]);
        }
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] *= scale;
    stv<CPT>(y + n * C + g * CPT, acc);
  }
}

// weight with which output index o reads input index i (exact transpose of lin_src)
__device__ __forceinline__ float lin_wt(int o, int i, float ratio, int n_in) {
  const Lin l = lin_src(o, ratio, n_in);
  return (l.i0 == i ? l.l0 : 0.f) + (l.i1 == i ? l.l1 : 0.f);
}
__device__ __forceinline__ void lin_range(int i, float ratio, int n_in, int& lo, int& hi) {
  const int n_out = 2 * n_in;
  if (ratio <= 0.f) { lo = 0; hi = n_out - 1; return; }
  lo = max(0, (int)floorf((float)(i - 1) / ratio) - 1);
  hi = min(n_out - 1, (int)ceilf((float)(i + 1) / ratio) + 1);
}

// gather form of the transposed interpolation: eight adjacent lanes share one (coarse voxel, channel group) and split
// the fine x range it touches (at most 5 voxels), so neighbouring lanes read neighbouring fine voxels and the small
// pyramid levels still fill the chip; fixed xor-tree over the eight lanes -> deterministic
template <int CPT>
__global__ __launch_bounds__(BLK) void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int d,
                                                            int h, int w, int C, int G, int64_t total, float scale) {
  const int H = 2 * h, W = 2 * w;
  const float rz = up_ratio(d), ry = up_ratio(h), rx = up_ratio(w);
  const int64_t Vo = (int64_t)8 * d * h * w, Vi = (int64_t)d * h * w;
  const int kx = threadIdx.x & 7;
  const int64_t stride = (int64_t)gridDim.x * (BLK / 8);
  // the eight lanes of a group share idx, so they enter and leave the loop together (the shuffles stay inside a group)
  for (int64_t idx = (int64_t)blockIdx.x * (BLK / 8) + (threadIdx.x >> 3); idx < total; idx += stride) {
    const int g = (int)(idx % G);
    const int64_t n = idx / G;
    const int64_t b = n / Vi, v = n - b * Vi;
    const int xi = (int)(v % w);
    const int64_t t2 = v / w;
    const int yi = (int)(t2 % h), zi = (int)(t2 / h);
    int zlo, zhi, ylo, yhi, xlo, xhi;
    lin_range(zi, rz, d, zlo, zhi);
    lin_range(yi, ry, h, ylo, yhi);
    lin_range(xi, rx, w, xlo, xhi);
    const float* gb = dy + b * Vo * C + g * CPT;
    float acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
    for (int xo = xlo + kx; xo <= xhi; xo += 8) {
      const float wx = lin_wt(xo, xi, rx, w);
      if (wx == 0.f) continue;
      for (int zo = zlo; zo <= zhi; ++zo) {
        const float wz = lin_wt(zo, zi, rz, d);
        if (wz == 0.f) continue;
        for (int yo = ylo; yo <= yhi; ++yo) {
          const float wy = lin_wt(yo, yi, ry, h);
          if (wy == 0.f) continue;
          float s[CPT];
          ldv<CPT>(gb + (((int64_t)zo * H + yo) * W + xo) * C, s);
          const float wgt = wz * wy * wx;
#pragma unroll
          for (int c = 0; c < CPT; ++c) acc[c] = fmaf(wgt, s[c], acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      float r = acc[c];
      r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
      acc[c] = r * scale;
    }
    if (kx == 0) stv<CPT>(dx + n * C + g * CPT, acc);
  }
}

// One axis of the same transposed interpolation: in [outer][2n][inner] -> out [outer][n][inner] (inner = everything
// faster than the axis, contiguous).  Three passes z, y, x replace the 3-D gather for the large levels: the first and
// largest pass reads whole (y,x) planes fully coalesced and every pass touches ~4 fine rows per output instead of the
// ~64 fine voxels of the direct form (level 1, C = 3: 147 us -> the three passes together).
// rows form (z and y passes: inner is thousands of floats): blockIdx.x = output row (o, i), so the candidate weights are
// computed once per workgroup with scalar arithmetic and the threads only stream `inner` (float4 when it divides)
template <bool V4>
__global__ __launch_bounds__(BLK) void upsample2_bwd_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int n,
                                                                 int64_t inner, float scale) {
  const float r = up_ratio(n);
  const int i = blockIdx.x % n;
  const int64_t o = blockIdx.x / n;
  int lo, hi;
  lin_range(i, r, n, lo, hi);
  while (lo < hi && lin_wt(lo, i, r, n) == 0.f) ++lo;      // (scalar) the conservative range starts with zero weights
  float wt[9];
#pragma unroll
  for (int a = 0; a < 9; ++a) wt[a] = lo + a <= hi ? lin_wt(lo + a, i, r, n) : 0.f;
  const int cnt = hi - lo + 1;                         // >= 1
  const float* p = in + (o * 2 * n + lo) * inner;
  float* q = out + (int64_t)blockIdx.x * inner;
  constexpr int E = V4 ? 4 : 1;
  for (int64_t k = ((int64_t)blockIdx.y * BLK + threadIdx.x) * E; k < inner; k += (int64_t)gridDim.y * BLK * E) {
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    // the first four candidates (all there are, except next to the borders) are loaded unconditionally and back to back
    // (a missing one re-reads the last valid row with weight 0); a load per `if (wt != 0)` was a round trip each
    float v4[4][E];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int aa = a < cnt ? a : cnt - 1;
      if (V4) {
        const float4 t = *reinterpret_cast<const float4*>(p + aa * inner + k);
        v4[a][0] = t.x; if (E > 1) { v4[a][1] = t.y; v4[a][2] = t.z; v4[a][3] = t.w; }
      } else {
        v4[a][0] = p[aa * inner + k];
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = fmaf(wt[a], v4[a][e], acc[e]);
#pragma unroll
    for (int a = 4; a < 9; ++a) {
      if (wt[a] == 0.f) continue;                      // uniform over the workgroup
      float v[E];
      if (V4) {
        const float4 t = *reinterpret_cast<const float4*>(p + a * inner + k);
        v[0] = t.x; if (E > 1) { v[1] = t.y; v[2] = t.z; v[3] = t.w; }
      } else {
        v[0] = p[a * inner + k];
      }
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = fmaf(wt[a], v[e], acc[e]);
    }
    if (V4) *reinterpret_cast<float4*>(q + k) = make_float4(acc[0] * scale, acc[E > 1 ? 1 : 0] * scale, acc[E > 1 ? 2 : 0] * scale,
                                                            acc[E > 1 ? 3 : 0] * scale);
    else q[k] = acc[0] * scale;
  }
}

// flat form (x pass: inner = C floats): one thread per output float, 32-bit index arithmetic
__global__ __launch_bounds__(BLK) void upsample2_bwd_axis_kernel(const float* __restrict__ in, float* __restrict__ out, int n,
                                                                 int inner, int64_t total, float scale) {
  const float r = up_ratio(n);
  for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
    const int64_t t = idx / inner;
    const int k = (int)(idx - t * inner);
    const int i = (int)(t % n);
    const int64_t o = t / n;
    int lo, hi;
    lin_range(i, r, n, lo, hi);
    const float* p = in + (o * 2 * n) * inner + k;
    float acc = 0.f;
    for (int f = lo; f <= hi; ++f) {
      const float wgt = lin_wt(f, i, r, n);
      if (wgt != 0.f) acc = fmaf(wgt, p[(int64_t)f * inner], acc);
    }
    out[idx] = acc * scale;
  }
}

// ------------------------------------------------------------------------------------------------ layout
// (B,C,V) -> (B,V,C): threads walk the output; reads are C strided streams, each coalesced across lanes
__global__ __launch_bounds__(BLK) void ncdhw_to_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                          int64_t V, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
    const int64_t b = idx / V, v = idx - b * V;
    for (int c = 0; c < C; ++c) y[idx * C + c] = x[(b * C + c) * V + v];
  }
}
__global__ __launch_bounds__(BLK) void cl_to_ncdhw_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                          int64_t V, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
    const int64_t b = idx / V, v = idx - b * V;
    for (int c = 0; c < C; ++c) y[(b * C + c) * V + v] = x[idx * C + c];
  }
}

// ------------------------------------------------------------------------------------------------ CWM tail
template <int HEADS>
__global__ __launch_bounds__(BLK) void cwm_tail_fwd_kernel(const float* __restrict__ x, const float* __restrict__ lg,
                                                           float* __restrict__ out, int64_t N) {
  for (int64_t n = (int64_t)blockIdx.x * BLK + threadIdx.x; n < N; n += (int64_t)gridDim.x * BLK) {
    float l[HEADS];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < HEADS; ++i) { l[i] = lg[n * HEADS + i]; m = fmaxf(m, l[i]); }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HEADS; ++i) { l[i] = __expf(l[i] - m); s += l[i]; }
    const float inv = 2.f / s;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int i = 0; i < HEADS; ++i) {
      const float* xp = x + n * (HEADS * 3) + i * 3;
      o0 = fmaf(l[i], xp[0], o0); o1 = fmaf(l[i], xp[1], o1); o2 = fmaf(l[i], xp[2], o2);
    }
    out[n * 3 + 0] = o0 * inv; out[n * 3 + 1] = o1 * inv; out[n * 3 + 2] = o2 * inv;
  }
}

template <int HEADS>
__global__ __launch_bounds__(BLK) void cwm_tail_bwd_kernel(const float* __restrict__ x, const float* __restrict__ lg,
                                                           const float* __restrict__ dout, float* __restrict__ dx,
                                                           float* __restrict__ dlg, int64_t N) {
  for (int64_t n = (int64_t)blockIdx.x * BLK + threadIdx.x; n < N; n += (int64_t)gridDim.x * BLK) {
    float p[HEADS], gdot[HEADS];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < HEADS; ++i) { p[i] = lg[n * HEADS + i]; m = fmaxf(m, p[i]); }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HEADS; ++i) { p[i] = __expf(p[i] - m); s += p[i]; }
    const float inv = 1.f / s;
    const float g0 = 2.f * dout[n * 3 + 0], g1 = 2.f * dout[n * 3 + 1], g2 = 2.f * dout[n * 3 + 2];
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < HEADS; ++i) {
      p[i] *= inv;
      const float* xp = x + n * (HEADS * 3) + i * 3;
      gdot[i] = g0 * xp[0] + g1 * xp[1] + g2 * xp[2];       // d out / d weight_i
      mean = fmaf(p[i], gdot[i], mean);
      float* dxp = dx + n * (HEADS * 3) + i * 3;
      dxp[0] = p[i] * g0; dxp[1] = p[i] * g1; dxp[2] = p[i] * g2;
    }
#pragma unroll
    for (int i = 0; i < HEADS; ++i) dlg[n * HEADS + i] = p[i] * (gdot[i] - mean);
  }
}

// ------------------------------------------------------------------------------------------------ eval tail
constexpr int MAXLAB = 256;
__global__ __launch_bounds__(BLK) void label_warp_counts_kernel(const int16_t* __restrict__ lm,
                                                                const float* __restrict__ flow,
                                                                const int16_t* __restrict__ lf,
                                                                int16_t* __restrict__ warped,
                                                                unsigned long long* __restrict__ counts, int D, int H,
                                                                int W, int nlab1) {
  __shared__ unsigned int hist[3 * MAXLAB];
  for (int i = threadIdx.x; i < 3 * nlab1; i += BLK) hist[i] = 0u;
  __syncthreads();
  const int64_t V = (int64_t)D * H * W;
  for (int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x; v < V; v += (int64_t)gridDim.x * BLK) {
    const int xi = (int)(v % W);
    const int64_t t2 = v / W;
    const int yi = (int)(t2 % H), zi = (int)(t2 / H);
    const float zr = rintf((float)zi + flow[v * 3 + 0]);
    const float yr = rintf((float)yi + flow[v * 3 + 1]);
    const float xr = rintf((float)xi + flow[v * 3 + 2]);
    int lab = 0;
    if (zr >= 0.f && zr < (float)D && yr >= 0.f && yr < (float)H && xr >= 0.f && xr < (float)W)
      lab = lm[((int64_t)zr * H + (int64_t)yr) * W + (int64_t)xr];
    if (warped) warped[v] = (int16_t)lab;
    const int tl = lf[v];
    if (lab >= 0 && lab < nlab1) atomicAdd(&hist[lab], 1u);
    if (tl >= 0 && tl < nlab1) atomicAdd(&hist[nlab1 + tl], 1u);
    if (lab == tl && lab >= 0 && lab < nlab1) atomicAdd(&hist[2 * nlab1 + lab], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * nlab1; i += BLK)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int pick_cpt(int C) { return (C % 4 == 0) ? 4 : (C % 3 == 0 ? 3 : 1); }

}  // namespace

#define DISPATCH_CPT(cpt, KERNEL, grid, stream, ...)                                                       \
  do {                                                                                                     \
    if ((cpt) == 4) hipLaunchKernelGGL(KERNEL<4>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__);          \
    else if ((cpt) == 3) hipLaunchKernelGGL(KERNEL<3>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__);     \
    else hipLaunchKernelGGL(KERNEL<1>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__);                     \
  } while (0)

extern "C" {

int modet_warp_fwd(const float* src, const float* flow, float* out, int B, int D, int H, int W, int C, int mode,
                   int add_flow, modet_stream_t stream) {
  MODET_CHECK_PTR(src); MODET_CHECK_PTR(flow); MODET_CHECK_PTR(out);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  if (mode != 0 && mode != 1) return MODET_ERR_UNSUPPORTED;
  if (add_flow && (C != 3 || mode != 0)) return MODET_ERR_DIM;
  const int cpt = pick_cpt(C), G = C / cpt;
  const int64_t total = (int64_t)B * D * H * W * G;
  if (total >= ((int64_t)1 << 31) - (int64_t)256 * 16 * BLK) return MODET_ERR_UNSUPPORTED;     // 32-bit item index
  DISPATCH_CPT(cpt, warp_fwd_kernel, flat_grid(total, BLK), (hipStream_t)stream, src, flow, out, D, H, W, C, G, total,
               mode, add_flow);
  return modet_launch_status();
}

#ifdef MODET_TUNING
int modet_debug_warp_census(unsigned long long* buf) {      // not in the header: tuning builds only
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wb_dbg), &buf, sizeof(buf));
}
#endif

int modet_warp_bwd(const float* src, const float* flow, const float* d_out, float* d_src, float* d_flow, int B, int D,
                   int H, int W, int C, int add_flow, int flow_bound, modet_stream_t stream) {
  return modet_warp_bwd_t(src, 0, flow, d_out, d_src, d_flow, B, D, H, W, C, add_flow, flow_bound, stream);
}

int modet_warp_bwd_t(const void* srcv, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow, int B,
                     int D, int H, int W, int C, int add_flow, int flow_bound, modet_stream_t stream) {
  return modet_warp_bwd_acc(srcv, src_bf16, flow, d_out, d_src, d_flow, nullptr, B, D, H, W, C, add_flow, flow_bound, stream);
}

int modet_warp_bwd_acc(const void* srcv, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow,
                       const float* d_flow_add, int B, int D, int H, int W, int C, int add_flow, int flow_bound,
                       modet_stream_t stream) {
  const float* src = (const float*)srcv;
  if (d_flow_add && (!d_flow || flow_bound != 0)) return MODET_ERR_UNSUPPORTED;
  MODET_CHECK_PTR(src); MODET_CHECK_PTR(flow); MODET_CHECK_PTR(d_out);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  if (add_flow && C != 3) return MODET_ERR_DIM;
  if (flow_bound != 0 && (flow_bound != 1 || C != 3)) return MODET_ERR_UNSUPPORTED;
  if (src_bf16 && (add_flow || flow_bound)) return MODET_ERR_UNSUPPORTED;      // (flow compositions stay fp32)
  if (!d_src && !d_flow) return MODET_OK;
  if (flow_bound == 1) {
    const int tx = cdiv(W, G3X), ty = cdiv(H, G3Y), tz = cdiv(D, G3Z);
    hipLaunchKernelGGL(warp_bwd_gather3_tiled_kernel, dim3(tx * ty * tz, B), dim3(BLK), 0, (hipStream_t)stream, src, flow,
                       d_out, d_src, d_flow, D, H, W, tx, ty, add_flow);
    return modet_launch_status();
  }
  int G = 1;
  while (G < C) G <<= 1;
  if (G > 64) return MODET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (d_src) modet_zero_async(d_src, (size_t)B * D * H * W * C * sizeof(float), s);     // (not hipMemsetAsync: common.h)
  // patch form (x, z AND y merges): 32-bit offsets, enough patch columns to fill the chip
  const int64_t total2 = (int64_t)B * cdiv(D, WB2_ZR) * cdiv(H, WB2_YR) * W * G;
  const char wb2 = modet_tuning_env("MODET_WARP_BWD2");
  if (d_src && wb2 != '0' && (int64_t)B * D * H * W * (C > 3 ? C : 3) * 4 < 0x7fffffffLL && (int64_t)(D + 3) * (H + 3) * (W + 3) < (1 << 23) &&
      W * C * 4 < (1 << 23) && total2 < 0x7fffffffLL && total2 >= 256 * 256) {
    if (src_bf16) hipLaunchKernelGGL((warp_bwd2_kernel<WB2_ZR, WB2_YR, true>), dim3(flat_grid(total2, BLK)), dim3(BLK), 0, s, src, flow,
                                     d_out, d_src, d_flow, d_flow_add, B, D, H, W, C, G, (unsigned)total2, add_flow);
    else hipLaunchKernelGGL((warp_bwd2_kernel<WB2_ZR, WB2_YR, false>), dim3(flat_grid(total2, BLK)), dim3(BLK), 0, s, src, flow, d_out,
                            d_src, d_flow, d_flow_add, B, D, H, W, C, G, (unsigned)total2, add_flow);
    return modet_launch_status();
  }
  const int64_t total = (int64_t)B * cdiv(D, ZRUN) * H * W * G;       // one item per (z run, y, x, channel slot)
  if (src_bf16) hipLaunchKernelGGL(warp_bwd_kernel<true>, dim3(flat_grid(total, BLK)), dim3(BLK), 0, s, src, flow, d_out, d_src, d_flow,
                                   d_flow_add, D, H, W, C, G, total, add_flow);
  else hipLaunchKernelGGL(warp_bwd_kernel<false>, dim3(flat_grid(total, BLK)), dim3(BLK), 0, s, src, flow, d_out, d_src, d_flow,
                          d_flow_add, D, H, W, C, G, total, add_flow);
  return modet_launch_status();
}

size_t modet_warp_bwd_det_ws_bytes(int B, int D, int H, int W, int C) {
  return 64 + (size_t)B * D * H * W * C * sizeof(long long);
}

int modet_warp_bwd_det(const void* srcv, int src_bf16, const float* flow, const float* d_out, float* d_src, float* d_flow,
                       const float* d_flow_add, void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, int add_flow,
                       modet_stream_t stream) {
  const float* src = (const float*)srcv;
  MODET_CHECK_PTR(src); MODET_CHECK_PTR(flow); MODET_CHECK_PTR(d_out); MODET_CHECK_PTR(d_src); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  if (add_flow && C != 3) return MODET_ERR_DIM;
  if (src_bf16 && add_flow) return MODET_ERR_UNSUPPORTED;
  if (d_flow_add && !d_flow) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_warp_bwd_det_ws_bytes(B, D, H, W, C)) return MODET_ERR_WORKSPACE;
  int G = 1;
  while (G < C) G <<= 1;
  if (G > 64) return MODET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)B * D * H * W * C;
  unsigned* hdr = (unsigned*)ws;
  long long* acc = (long long*)((char*)ws + 64);
  modet_zero_async(ws, 64 + (size_t)n * sizeof(long long), s);
  hipLaunchKernelGGL(absmax_kernel, dim3(flat_grid(n, BLK)), dim3(BLK), 0, s, d_out, n, hdr);
  const int64_t total2 = (int64_t)B * cdiv(D, WB2_ZR) * cdiv(H, WB2_YR) * W * G;
  if ((int64_t)B * D * H * W * (C > 3 ? C : 3) * 4 < 0x7fffffffLL && (int64_t)(D + 3) * (H + 3) * (W + 3) < (1 << 23) &&
      W * C * 4 < (1 << 23) && total2 < 0x7fffffffLL && total2 >= 256 * 256) {
    if (src_bf16) hipLaunchKernelGGL((warp_bwd2_kernel<WB2_ZR, WB2_YR, true, true>), dim3(flat_grid(total2, BLK)), dim3(BLK), 0, s, src,
                                     flow, d_out, d_src, d_flow, d_flow_add, B, D, H, W, C, G, (unsigned)total2, add_flow, acc, hdr);
    else hipLaunchKernelGGL((warp_bwd2_kernel<WB2_ZR, WB2_YR, false, true>), dim3(flat_grid(total2, BLK)), dim3(BLK), 0, s, src, flow,
                            d_out, d_src, d_flow, d_flow_add, B, D, H, W, C, G, (unsigned)total2, add_flow, acc, hdr);
  } else {
    const int64_t total = (int64_t)B * cdiv(D, ZRUN) * H * W * G;
    if (src_bf16) hipLaunchKernelGGL((warp_bwd_kernel<true, true>), dim3(flat_grid(total, BLK)), dim3(BLK), 0, s, src, flow, d_out, d_src,
                                     d_flow, d_flow_add, D, H, W, C, G, total, add_flow, acc, hdr);
    else hipLaunchKernelGGL((warp_bwd_kernel<false, true>), dim3(flat_grid(total, BLK)), dim3(BLK), 0, s, src, flow, d_out, d_src,
                            d_flow, d_flow_add, D, H, W, C, G, total, add_flow, acc, hdr);
  }
  hipLaunchKernelGGL(det_decode_kernel, dim3(flat_grid(n, BLK)), dim3(BLK), 0, s, (const long long*)acc, d_src, n, (const unsigned*)hdr);
  return modet_launch_status();
}

int modet_warp_fwd_t(const void* src, int src_bf16, const float* flow, void* out, int out_bf16, int B, int D, int H, int W, int C,
                     modet_stream_t stream) {
  if (!src_bf16 && !out_bf16) return modet_warp_fwd((const float*)src, flow, (float*)out, B, D, H, W, C, 0, 0, stream);
  MODET_CHECK_PTR(src); MODET_CHECK_PTR(flow); MODET_CHECK_PTR(out);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  if (C % 4 != 0) return MODET_ERR_UNSUPPORTED;
  const int G = C / 4;
  const int64_t total = (int64_t)B * D * H * W * G;
  if (total >= ((int64_t)1 << 31) - (int64_t)256 * 16 * BLK) return MODET_ERR_UNSUPPORTED;     // 32-bit item index
  const dim3 grid(flat_grid(total, BLK));
  hipStream_t s = (hipStream_t)stream;
  const float* sf = (const float*)src;
  float* of = (float*)out;
  if (src_bf16 && out_bf16) hipLaunchKernelGGL((warp_fwd_kernel<4, true, true>), grid, dim3(BLK), 0, s, sf, flow, of, D, H, W, C, G, total, 0, 0);
  else if (src_bf16) hipLaunchKernelGGL((warp_fwd_kernel<4, false, true>), grid, dim3(BLK), 0, s, sf, flow, of, D, H, W, C, G, total, 0, 0);
  else hipLaunchKernelGGL((warp_fwd_kernel<4, true, false>), grid, dim3(BLK), 0, s, sf, flow, of, D, H, W, C, G, total, 0, 0);
  return modet_launch_status();
}

int modet_warp_fwd_o16(const float* src, const float* flow, void* out_bf16, int B, int D, int H, int W, int C,
                       modet_stream_t stream) {
  return modet_warp_fwd_t(src, 0, flow, out_bf16, 1, B, D, H, W, C, stream);
}

int modet_upsample2_fwd(const float* x, float* y, int B, int d, int h, int w, int C, float scale,
                        modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(B > 0 && d > 0 && h > 0 && w > 0 && C > 0);
  const int cpt = pick_cpt(C), G = C / cpt;
  const int64_t total = (int64_t)B * 8 * d * h * w * G;
  DISPATCH_CPT(cpt, upsample2_fwd_kernel, flat_grid(total, BLK), (hipStream_t)stream, x, y, d, h, w, C, G, total,
               scale);
  return modet_launch_status();
}

int modet_upsample2_bwd(const float* d_y, float* d_x, int B, int d, int h, int w, int C, float scale,
                        modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(d_x);
  MODET_CHECK_DIM(B > 0 && d > 0 && h > 0 && w > 0 && C > 0);
  const int cpt = pick_cpt(C), G = C / cpt;
  const int64_t total = (int64_t)B * d * h * w * G;
  DISPATCH_CPT(cpt, upsample2_bwd_kernel, flat_grid(total * 8, BLK), (hipStream_t)stream, d_y, d_x, d, h, w, C, G, total,
               scale);
  return modet_launch_status();
}

size_t modet_upsample2_bwd_sep_ws_bytes(int B, int d, int h, int w, int C) {
  if ((int64_t)B * d * h * w < 65536) return 0;            // small levels: one launch of the direct gather is cheaper
  return ((size_t)B * d * 2 * h * 2 * w * C + (size_t)B * d * h * 2 * w * C) * sizeof(float);
}

int modet_upsample2_bwd_sep(const float* d_y, float* d_x, void* ws, size_t ws_bytes, int B, int d, int h, int w, int C,
                            float scale, modet_stream_t stream) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && d > 0 && h > 0 && w > 0 && C > 0);
  const size_t need = modet_upsample2_bwd_sep_ws_bytes(B, d, h, w, C);
  if (need == 0) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < need) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* t1 = (float*)ws;
  float* t2 = t1 + (size_t)B * d * 2 * h * 2 * w * C;
  const int64_t in_z = (int64_t)4 * h * w * C, in_y = (int64_t)2 * w * C;
  const int64_t n1 = (int64_t)B * d * in_z, n2 = (int64_t)B * d * h * in_y, n3 = (int64_t)B * d * h * w * C;
  (void)n1; (void)n2;
  auto rows = [&](const float* in, float* out, int n, int64_t nrows, int64_t inner) {
    int gy = (int)cdiv64(inner, (int64_t)BLK * 4);
    if (gy > 64) gy = 64;
    if (inner % 4 == 0)
      hipLaunchKernelGGL(upsample2_bwd_rows_kernel<true>, dim3((unsigned)nrows, gy), dim3(BLK), 0, s, in, out, n, inner, 1.f);
    else
      hipLaunchKernelGGL(upsample2_bwd_rows_kernel<false>, dim3((unsigned)nrows, gy), dim3(BLK), 0, s, in, out, n, inner, 1.f);
  };
  rows(d_y, t1, d, (int64_t)B * d, in_z);
  rows(t1, t2, h, (int64_t)B * d * h, in_y);
  hipLaunchKernelGGL(upsample2_bwd_axis_kernel, dim3(flat_grid(n3, BLK)), dim3(BLK), 0, s, (const float*)t2, d_x, w, C, n3, scale);
  return modet_launch_status();
}

int modet_ncdhw_to_cl(const float* x, float* y, int B, int C, int64_t V, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(B > 0 && C > 0 && V > 0);
  const int64_t total = (int64_t)B * V;
  hipLaunchKernelGGL(ncdhw_to_cl_kernel, dim3(flat_grid(total, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, y, C, V,
                     total);
  return modet_launch_status();
}

int modet_cl_to_ncdhw(const float* x, float* y, int B, int C, int64_t V, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(y);
  MODET_CHECK_DIM(B > 0 && C > 0 && V > 0);
  const int64_t total = (int64_t)B * V;
  hipLaunchKernelGGL(cl_to_ncdhw_kernel, dim3(flat_grid(total, BLK)), dim3(BLK), 0, (hipStream_t)stream, x, y, C, V,
                     total);
  return modet_launch_status();
}

#define DISPATCH_HEADS(KERNEL, ...)                                                                               \
  switch (heads) {                                                                                                \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(flat_grid(N, BLK)), dim3(BLK), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(flat_grid(N, BLK)), dim3(BLK), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(flat_grid(N, BLK)), dim3(BLK), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(flat_grid(N, BLK)), dim3(BLK), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    default: return MODET_ERR_UNSUPPORTED;                                                                        \
  }

int modet_cwm_tail_fwd(const float* x, const float* logits, float* out, int64_t N, int heads, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(logits); MODET_CHECK_PTR(out);
  MODET_CHECK_DIM(N > 0 && heads > 0);
  DISPATCH_HEADS(cwm_tail_fwd_kernel, x, logits, out, N);
  return modet_launch_status();
}

int modet_cwm_tail_bwd(const float* x, const float* logits, const float* d_out, float* d_x, float* d_logits,
                       int64_t N, int heads, modet_stream_t stream) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(logits); MODET_CHECK_PTR(d_out); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(d_logits);
  MODET_CHECK_DIM(N > 0 && heads > 0);
  DISPATCH_HEADS(cwm_tail_bwd_kernel, x, logits, d_out, d_x, d_logits, N);
  return modet_launch_status();
}

int modet_label_warp_counts(const int16_t* lab_moving, const float* flow, const int16_t* lab_fixed, int16_t* warped,
                            int64_t* counts, int D, int H, int W, int nlabels, modet_stream_t stream) {
  MODET_CHECK_PTR(lab_moving); MODET_CHECK_PTR(flow); MODET_CHECK_PTR(lab_fixed); MODET_CHECK_PTR(counts);
  MODET_CHECK_DIM(D > 0 && H > 0 && W > 0 && nlabels > 0);
  if (nlabels + 1 > MAXLAB) return MODET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  modet_zero_async(counts, (size_t)3 * (nlabels + 1) * sizeof(int64_t), s);
  const int64_t V = (int64_t)D * H * W;
  int grid = flat_grid(V, BLK);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(label_warp_counts_kernel, dim3(grid), dim3(BLK), 0, s, lab_moving, flow, lab_fixed, warped,
                     (unsigned long long*)counts, D, H, W, nlabels + 1);
  return modet_launch_status();
}

}  // extern "C"
