// 3x3x3 / stride 1 / zero-pad 1 convolution with bf16 STORAGE and fp32 ACCUMULATION on gfx950's bf16 matrix pipe
// (v_mfma_f32_16x16x32_bf16: 16x the rate of the exact-f32 MFMA the fp32 path uses) -- BASELINE.json configs[4]
// ("bf16 storage / fp32 accumulate").  Channels-last activations; fp32 master weights are packed to bf16 per launch;
// bias, InstanceNorm statistics and weight gradients stay fp32.
//   reference call sites: nn.Conv3d in ConvInsBlock, ModeT/models.py:135-151 (arithmetic lives in ATen/MIOpen there;
//   the reference itself has no reduced-precision path: cfg 5 is this build's own configuration).
//
// forward / dgrad:  D[voxel(16 along W)][cout(16)] += A[voxel][k(32)] * B[k(32)][cout],  k = (tap, cin) with cin fastest.
//   A: one ds_read_b128 per lane = 8 consecutive input channels of one tap at one voxel, from an LDS tile
//      [halo'd voxel][CK channels] (bf16; fp32 inputs are converted while the tile is staged, zero padding included);
//   B: one 16-byte global load per lane from weights packed [stage][k-step][cout][32 k] (L2-resident, no LDS copy),
//      re-used by all the M tiles (output rows) of the wave;
//   the MFMA takes the weights as its A and the voxels as its B operand, so a lane ends up with 4 consecutive couts
//      of one voxel; epilogue: + bias, fused InstanceNorm statistics from the fp32 accumulators (shifted sums, one row
//      per workgroup, same buffer format as the fp32 path: see ConvIn in conv3d.hip), direct 8/16-byte stores.
// One workgroup = one output tile (no persistence: the tile's arithmetic is a few hundred MFMAs, the kernel is bound by
// staging the halo'd tile; 3-5 workgroups per CU overlap each other's phases).
// dgrad is the same kernel on flipped + transposed weights.
#include "common.h"
#include "step_ctx.h"
#include <mutex>
#include <vector>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 256;
constexpr int TX = 16, HX = TX + 2;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {        // round to nearest even (v_cvt_pk_bf16_f32)
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ unsigned short to_bf16(float a) {
  const __bf16 x = (__bf16)a;
  return __builtin_bit_cast(unsigned short, x);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
// x = hi + mid + lo exactly to 2^-24 relative (three 8-bit significands, each piece rounded to nearest even; the
// remainders x - hi and (x - hi) - mid are exact in fp32)
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = to_bf16(x);
  const float r1 = x - bf16_to_f32(hi);
  mid = to_bf16(r1);
  const float r2 = r1 - bf16_to_f32(mid);
  lo = to_bf16(r2);
}

// ------------------------------------------------------------------------------------------------ weight packing
// wpk[stage][step][n < CoutP][32] bf16: element j of k-step `step` is k' = step*32 + j within the stage,
//   tap = k' / CK, channel c = stage*CK + k' % CK; zero for tap >= 27, c >= Cin, n >= Cout.
//   mode 0 (forward): w[n][c][tap]   (w: (Cout, Cin, 27));   mode 1 (dgrad): w[c][n][26 - tap]  (w: (Co = c, Ci = n, 27))
//   npiece = 3 (fp32 emulation, see conv3d_bf16_kernel SP = 3): three such arrays back to back holding the hi / mid / lo
//   bf16 pieces of every weight.
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cin, int Cout,
                                         int CoutP, int CK, int nstage, int ksteps, int mode, int npiece) {
  const int total = nstage * ksteps * CoutP * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 31;
    int t = i >> 5;
    const int n = t % CoutP; t /= CoutP;
    const int step = t % ksteps, stage = t / ksteps;
    const int kp = step * 32 + j;
    const int tap = kp / CK, c = stage * CK + kp % CK;
    float v = 0.f;
    if (tap < 27 && c < Cin && n < Cout)
      v = mode == 0 ? w[((int64_t)n * Cin + c) * 27 + tap] : w[((int64_t)c * Cout + n) * 27 + 26 - tap];
    if (npiece == 1) {
      wpk[i] = to_bf16(v);
    } else {
      unsigned short h, m, l;
      split3(v, h, m, l);
      wpk[i] = h; wpk[(size_t)total + i] = m; wpk[(size_t)2 * total + i] = l;
    }
  }
}

// The same packing for many weight tensors in one launch (modet_conv3d_prepack_*, see conv3d.hip): jobs by value.
constexpr int PACKB_MAX_JOBS = 40;
struct PackBJob { const float* w; unsigned short* wpk; int Cin, Cout, CoutP, CK, nstage, ksteps, mode, npiece; };
struct PackBTable { PackBJob job[PACKB_MAX_JOBS]; int n; };
__global__ void pack_weights_bf16_many_kernel(const PackBTable t) {
  const PackBJob& J = t.job[blockIdx.y];
  const float* __restrict__ w = J.w;
  unsigned short* __restrict__ wpk = J.wpk;
  const int Cin = J.Cin, Cout = J.Cout, CoutP = J.CoutP, CK = J.CK, ksteps = J.ksteps, mode = J.mode;
  const int total = J.nstage * ksteps * CoutP * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 31;
    int tt = i >> 5;
    const int n = tt % CoutP; tt /= CoutP;
    const int step = tt % ksteps, stage = tt / ksteps;
    const int kp = step * 32 + j;
    const int tap = kp / CK, c = stage * CK + kp % CK;
    float v = 0.f;
    if (tap < 27 && c < Cin && n < Cout)
      v = mode == 0 ? w[((int64_t)n * Cin + c) * 27 + tap] : w[((int64_t)c * Cout + n) * 27 + 26 - tap];
    if (J.npiece == 1) {
      wpk[i] = to_bf16(v);
    } else {
      unsigned short h, m, l;
      split3(v, h, m, l);
      wpk[i] = h; wpk[(size_t)total + i] = m; wpk[(size_t)2 * total + i] = l;
    }
  }
}

inline size_t packb_elems(const PackBKey& k) {
  return ((size_t)k.nstage * k.ksteps * k.CoutP * 32 * k.npiece + 127) / 128 * 128;
}
const unsigned short* prepacked_bf16_or_record(modet_step_ctx* c, const PackBKey& k) {
  if (!c) return nullptr;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->active) {
    for (size_t i = 0; i < c->bjobs.size(); ++i)
      if (c->bjobs[i] == k) return c->barena + c->boff[i];
    return nullptr;
  }
  if (c->recording) {
    bool seen = false;
    for (const PackBKey& j : c->bjobs) seen = seen || j == k;
    if (!seen) c->bjobs.push_back(k);
  }
  return nullptr;
}

// shift K of the fused statistics (see conv_shift_kernel in conv3d.hip): the conv output at voxel (1,1,1), any summation order
template <bool IN_BF16>
__global__ __launch_bounds__(256) void conv_shift_bf16_kernel(const void* __restrict__ xv, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ shift,
                                                              int B, int D, int H, int W, int Cin, int Cout) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= B * Cout) return;
  const int b = item / Cout, co = item - b * Cout;
  const int z = D > 1 ? 1 : 0, y = H > 1 ? 1 : 0, xx = W > 1 ? 1 : 0;
  float acc = 0.f;
  const int n = 27 * Cin;
  const int64_t xb = (int64_t)b * D * H * W * Cin;
  // four (tap, cin) items per lane and trip with their loads in flight together (see conv_shift_kernel in conv3d.hip)
  for (int i0 = lane; i0 < n; i0 += 256) {
    unsigned xr[4];
    float wv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 64 * u;
      const int ii = i < n ? i : 0;
      const int tap = ii / Cin, c = ii - tap * Cin;
      const int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, xq = xx + tap % 3 - 1;
      ok[u] = i < n && zz >= 0 && zz < D && yy >= 0 && yy < H && xq >= 0 && xq < W;
      const int64_t off = xb + (ok[u] ? (((int64_t)zz * H + yy) * W + xq) * Cin + c : 0);
      xr[u] = IN_BF16 ? (unsigned)((const unsigned short*)xv)[off] : __float_as_uint(((const float*)xv)[off]);    // raw
      wv[u] = w[((int64_t)co * Cin + c) * 27 + tap];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(xr[u]), "+v"(wv[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = IN_BF16 ? bf16_to_f32((unsigned short)xr[u]) : __uint_as_float(xr[u]);
      if (ok[u]) acc = fmaf(v, wv[u], acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) shift[item] = acc + (bias ? bias[co] : 0.f);
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
// SP = 3: fp32 EMULATION on the bf16 pipe ("bf16x3").  Inputs and weights are fp32; each is split into three bf16 pieces
// (hi + mid + lo = the fp32 value to 2^-24) while it is staged, and every product a*b is evaluated as the six piece
// products of total order <= 2 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi -- small terms first), each EXACT in the
// fp32 accumulator: the three dropped terms are <= 3 * 2^-24 |a b|, the same class as the single rounding of an fp32 FMA.
// Six bf16 MFMAs replace 8 passes of the exact-f32 MFMA at 1/16 of the rate: 2.7x less matrix-pipe time at fp32
// accuracy (outputs, statistics and the tensors in HBM stay fp32).
template <int TZ, int TY, int CK, int NT, bool IN_BF16, bool OUT_BF16, bool STATS, int SP = 1>
__global__ __launch_bounds__(NTHR) void conv3d_bf16_kernel(const void* __restrict__ xin, const uint4* __restrict__ wpk,
                                                           const float* __restrict__ bias, void* __restrict__ yout,
                                                           float* __restrict__ stats_rows, const float* __restrict__ shift,
                                                           int D, int H, int W, int Cin, int Cout, int CoutP, int nstage,
                                                           int tiles_x, int tiles_y, int wpiece) {
  constexpr int HZ = TZ + 2, HY = TY + 2, HVOX = HZ * HY * HX;
  constexpr int ROWS = TZ * TY, RW = ROWS / 4;            // output rows (M tiles of 16 voxels) per workgroup / per wave
  constexpr int NCB = NT * 16;
  constexpr int KSTEPS = (27 * CK + 31) / 32;
  constexpr int CKB = CK / 8;                             // 16-byte channel blocks per voxel in the LDS tile
  constexpr int PLANE = HVOX * CK * 2;                    // bytes of one piece's tile
  constexpr int LDS_BYTES = PLANE * SP;
  static_assert(ROWS % 4 == 0, "rows split over 4 waves");
  static_assert(SP == 1 || (SP == 3 && !IN_BF16), "the fp32 emulation takes fp32 inputs");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  __shared__ float sred[STATS ? 4 * NCB * 2 : 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int cb0 = blockIdx.y * NCB, b = blockIdx.z;
  int t = blockIdx.x;
  const int x0 = (t % tiles_x) * TX; t /= tiles_x;
  const int y0 = (t % tiles_y) * TY;
  const int z0 = (t / tiles_y) * TZ;
  const int64_t vbase = (int64_t)b * D * H * W;

  f32x4 acc[RW][NT];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int rowoff[RW];                                          // element offset of (row, voxel li) in the LDS tile, tap (0,0,0)
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int rr = wave * RW + r;
    rowoff[r] = (((rr / TY) * HY + (rr % TY)) * HX + li) * CK;
  }

  for (int s = 0; s < nstage; ++s) {
    if (s > 0) __syncthreads();                            // every wave is done reading the previous stage's tile
    // ---- stage the halo'd input tile: [voxel][CK] bf16, zeros outside the volume and beyond Cin.  All global loads of a
    // batch are issued before the first LDS write (a load -> write -> load chain exposed one memory round trip per item:
    // 7-15 per tile, 24 us per tile against 1 us of MFMA work).
    constexpr int NITEM = HVOX * CKB, NIT = (NITEM + NTHR - 1) / NTHR;
    constexpr int BATCH = IN_BF16 ? 8 : 4;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
      uint4 raw[BATCH];
      float4 rf[IN_BF16 ? 1 : BATCH][2];
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        const int idx = tid + (it0 + q) * NTHR;
        raw[q] = make_uint4(0u, 0u, 0u, 0u);
        if (!IN_BF16) { rf[q][0] = make_float4(0.f, 0.f, 0.f, 0.f); rf[q][1] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (it0 + q < NIT && idx < NITEM) {
          const int hv = idx / CKB, cb = idx - hv * CKB;
          const int hx = hv % HX, t2 = hv / HX;
          const int hy = t2 % HY, hz = t2 / HY;
          const int z = z0 + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
          const int c = s * CK + cb * 8;
          if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W && c < Cin) {
            const int64_t off = (vbase + ((int64_t)z * H + yy) * W + xx) * Cin + c;
            if (IN_BF16) {
              raw[q] = *reinterpret_cast<const uint4*>((const unsigned short*)xin + off);       // Cin % 8 == 0 (host checks)
            } else {
              const float* p = (const float*)xin + off;
              rf[q][0] = *reinterpret_cast<const float4*>(p);                                   // Cin % 4 == 0 (host checks)
              if (c + 4 < Cin) rf[q][1] = *reinterpret_cast<const float4*>(p + 4);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        const int idx = tid + (it0 + q) * NTHR;
        if (it0 + q < NIT && idx < NITEM) {
          const int hv = idx / CKB, cb = idx - hv * CKB;
          unsigned char* dst = lds + ((size_t)hv * CK + cb * 8) * 2;
          if constexpr (SP == 3) {
            const float f[8] = {rf[q][0].x, rf[q][0].y, rf[q][0].z, rf[q][0].w, rf[q][1].x, rf[q][1].y, rf[q][1].z, rf[q][1].w};
            unsigned pw[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              unsigned short h0, m0, l0, h1, m1, l1;
              split3(f[2 * e], h0, m0, l0);
              split3(f[2 * e + 1], h1, m1, l1);
              pw[0][e] = (unsigned)h0 | ((unsigned)h1 << 16);
              pw[1][e] = (unsigned)m0 | ((unsigned)m1 << 16);
              pw[2][e] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              *reinterpret_cast<uint4*>(dst + (size_t)pc * PLANE) = make_uint4(pw[pc][0], pw[pc][1], pw[pc][2], pw[pc][3]);
          } else {
            uint4 v = raw[q];
            if (!IN_BF16) {
              v.x = pack_bf16x2(rf[q][0].x, rf[q][0].y); v.y = pack_bf16x2(rf[q][0].z, rf[q][0].w);
              v.z = pack_bf16x2(rf[q][1].x, rf[q][1].y); v.w = pack_bf16x2(rf[q][1].z, rf[q][1].w);
            }
            *reinterpret_cast<uint4*>(dst) = v;
          }
        }
      }
    }
    __syncthreads();
    // ---- K loop: k-step = 32 k values = 4 lane groups x 8 consecutive channels of one tap
    const uint4* wst = wpk + ((size_t)s * KSTEPS * CoutP + cb0) * 4;
    // a RUN-TIME loop over the k-steps (fully unrolled, the compiler hoisted the LDS reads of all 7..27 steps and the
    // kernel needed 244..418 registers: 1-2 waves per SIMD, latency-bound); the next step's weights are prefetched
    uint4 bcur[SP][NT], bnext[SP][NT];
#pragma unroll
    for (int pc = 0; pc < SP; ++pc)
#pragma unroll
      for (int n = 0; n < NT; ++n) bcur[pc][n] = wst[(size_t)pc * wpiece + (size_t)(n * 16 + li) * 4 + lk];
#pragma unroll 1
    for (int step = 0; step < KSTEPS; ++step) {
      if (step + 1 < KSTEPS) {
#pragma unroll
        for (int pc = 0; pc < SP; ++pc)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            bnext[pc][n] = wst[(size_t)pc * wpiece + ((size_t)(step + 1) * CoutP + n * 16 + li) * 4 + lk];
      }
      const int kb = step * 4 + lk;                        // this lane group's block of 8 k values
      int tap = (kb * 8) / CK;
      const int c0 = (kb * 8) % CK;
      if (tap > 26) tap = 26;                              // padded k: finite data times zero weights
      const int toff = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK + c0;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        bf16x8 a[SP];
#pragma unroll
        for (int pc = 0; pc < SP; ++pc)
          a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (size_t)pc * PLANE + (size_t)(rowoff[r] + toff) * 2));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#define MM(WP, AP) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bcur[WP][n]), a[AP], acc[r][n], 0, 0, 0)
          if constexpr (SP == 3) { MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0); }
          else MM(0, 0);
#undef MM
        }
      }
#pragma unroll
      for (int pc = 0; pc < SP; ++pc)
#pragma unroll
        for (int n = 0; n < NT; ++n) bcur[pc][n] = bnext[pc][n];
    }
  }

  // ---- epilogue.  The MFMA ran with the operands swapped (A = weights, B = voxels), so D is [cout][voxel]: this lane
  // holds voxel x = li and the 4 CONSECUTIVE couts 4*lk .. 4*lk+3 of each n tile: bias, statistics and the store need no
  // transpose -- one 8-byte (bf16) or 16-byte (fp32) store per (row, n tile), 16 voxels x 32 contiguous bytes per wave.
  constexpr int OSZ = OUT_BF16 ? 2 : 4;
  float sx[NT][4], sq[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) { sx[n][j] = 0.f; sq[n][j] = 0.f; }
  const bool xin_ok = x0 + li < W;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = cb0 + n * 16 + lk * 4;                  // first of this lane's 4 couts (Cout % 4 == 0)
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, kv[4] = {0.f, 0.f, 0.f, 0.f};
    if (co < Cout) {
      if (bias) { const float4 t4 = *reinterpret_cast<const float4*>(bias + co); bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w; }
      if (STATS) { const float4 t4 = *reinterpret_cast<const float4*>(shift + b * Cout + co); kv[0] = t4.x; kv[1] = t4.y; kv[2] = t4.z; kv[3] = t4.w; }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int rr = wave * RW + r;
      const int z = z0 + rr / TY, yy = y0 + rr % TY;
      const bool ok = xin_ok && z < D && yy < H && co < Cout;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[r][n][j] + bv[j];
        if (STATS && ok) { const float e = v[j] - kv[j]; sx[n][j] += e; sq[n][j] = fmaf(e, e, sq[n][j]); }
      }
      if (ok) {
        const int64_t off = (vbase + ((int64_t)z * H + yy) * W + x0 + li) * Cout + co;
        if (OUT_BF16) *reinterpret_cast<uint2*>((unsigned short*)yout + off) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        else *reinterpret_cast<float4*>((float*)yout + off) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (STATS) {
    // sum over the 16 voxel lanes (li), then over the 4 waves through LDS: one row per (sample, tile)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { sx[n][j] += __shfl_xor(sx[n][j], o, 64); sq[n][j] += __shfl_xor(sq[n][j], o, 64); }
        if (li == 0) {
          sred[(wave * NCB + n * 16 + lk * 4 + j) * 2] = sx[n][j];
          sred[(wave * NCB + n * 16 + lk * 4 + j) * 2 + 1] = sq[n][j];
        }
      }
    __syncthreads();
    if (tid < NCB * 2) {
      const int col = tid >> 1, which = tid & 1;
      if (cb0 + col < Cout) {
        float a = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) a += sred[(w4 * NCB + col) * 2 + which];
        stats_rows[(((int64_t)b * gridDim.x + blockIdx.x) * Cout + cb0 + col) * 2 + which] = a;    // [b][tile][Cout][2]
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad
// d_w[co][ci][tap] = sum_v d_y[v][co] * x[v + off(tap)][ci]  as  D[m][n] += A[m][k] B[k][n]  with  k = 32 voxels along W,
// m = (tap, ci) in tiles of 16 rows, n = cout.  Both operands need 8 CONSECUTIVE VOXELS of one channel per lane, so the
// tiles are staged channel-planar in LDS (the global reads stay channels-last; two x-adjacent voxels are packed per
// ds_write_b32):  xs[ci][z][y][48] with the tile's x origin at element 8, dys[co][row][32].
// The three x taps of one (dz, dy, ci) row come from ONE set of aligned reads: with P, Q, R the 16-byte blocks at
// elements 8*lk, 8*lk+8, 8*lk+16, the runs starting at elements 8*lk+7 / +8 / +9 (dx = 0 / 1 / 2) are funnel shifts
// (v_alignbit) of {P.w, Q, R.x}.  An M tile is 16 rows sharing one dx: 16 channels of one (dz,dy) (CIB = 16), 2 (dz,dy)
// combinations x 8 channels (CIB = 8) or 4 x 4 (CIB = 4); a "group" = the 3 M tiles (dx) built from one set of reads.
// Workgroups are persistent over voxel tiles, each wave takes every 4th row (k-step) of a tile and keeps private fp32
// accumulators for all tiles; at the end the 4 waves are summed through LDS and the workgroup writes ONE partial per
// (group, dx, n tile), which wgrad_bf16_reduce_kernel sums over workgroups in fixed order in fp64 (deterministic).
// Slot U*3 (row 0 of an all-ones A) carries d_bias.
constexpr int WTX = 32, WHXP = 48;
template <int CIB, int U, int NTB, int TZ, int TY, bool X_BF16>
__global__ __launch_bounds__(NTHR) void conv3d_bf16_wgrad_kernel(const void* __restrict__ xin, const unsigned short* __restrict__ dy,
                                                                 float* __restrict__ part, int D, int H, int W, int Cin, int Cout,
                                                                 int tiles_x, int tiles_y, int tiles_z, int ntiles, int n_coblk) {
  constexpr int HZ = TZ + 2, HY = TY + 2, ROWS = TZ * TY;
  constexpr int PX = HZ * HY * WHXP + 8;                    // plane stride (elements); +8 spreads the channel planes over the banks
  constexpr int PD = ROWS * WTX + 8;
  constexpr int NCO = NTB * 16;
  constexpr int SLOTS = U * 3 + 1;
  constexpr int XS_EL = CIB * PX, DS_EL = NCO * PD;
  constexpr int RED_FL = SLOTS * NTB * 256;                 // floats of the cross-wave reduction buffer
  constexpr int LDS_BYTES = ((XS_EL + DS_EL) * 2 > RED_FL * 4 ? (XS_EL + DS_EL) * 2 : RED_FL * 4);
  static_assert(ROWS % 4 == 0, "rows split over 4 waves");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned short* xs = reinterpret_cast<unsigned short*>(lds);
  unsigned short* dys = xs + XS_EL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int chunk = blockIdx.y / n_coblk, yb = blockIdx.y - chunk * n_coblk;   // group chunk, cout block
  const int co0 = yb * NCO;
  // this workgroup's channel block and first group
  constexpr int GPB = CIB == 16 ? 9 : (CIB == 8 ? 5 : 3);    // groups per channel block
  const int g0 = chunk * U;                                   // global group index of local group 0
  const int cblk = g0 / GPB;                                  // (U divides GPB, so a chunk never straddles blocks)
  const int ci0 = cblk * CIB;
  const bool do_bias = chunk == 0;

  // per-lane A row: channel and the (dz,dy) combination of each local group
  const int myci = CIB == 16 ? li : (CIB == 8 ? (li & 7) : (li & 3));
  int aoff[U];                                                // element offset of (ci plane, dz, dy, 8*lk) for row (0,0)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int gl = (g0 + u) % GPB;
    int combo = CIB == 16 ? gl : (CIB == 8 ? 2 * gl + (li >> 3) : 4 * gl + (li >> 2));
    if (combo > 8) combo = 8;                                 // dummy rows: finite data, never written out
    aoff[u] = myci * PX + ((combo / 3) * HY + combo % 3) * WHXP + 8 * lk;
  }

  f32x4 acc[U][3][NTB];
  f32x4 accb[NTB];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int n = 0; n < NTB; ++n) acc[u][d][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < NTB; ++n) accb[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned one2 = li == 0 ? 0x3f803f80u : 0u;           // bf16 (1, 1): row 0 of the bias tile
  const uint4 ones = make_uint4(one2, one2, one2, one2);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int x0 = (t % tiles_x) * WTX; t /= tiles_x;
    const int y0 = (t % tiles_y) * TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * TZ;
    const int64_t vbase = (int64_t)(t / tiles_z) * D * H * W;
    __syncthreads();                                          // every wave is done with the previous tile
    // ---- x tile: elements 6..41 of each (z,y) row (volume x = x0 - 8 + e), 18 voxel pairs per row
    for (int idx = tid; idx < HZ * HY * 18; idx += NTHR) {
      const int p = idx % 18 + 3, r2 = idx / 18;
      const int hy = r2 % HY, hz = r2 / HY;
      const int z = z0 + hz - 1, yy = y0 + hy - 1, xx = x0 - 8 + 2 * p;
      float v0[CIB], v1[CIB];
      unsigned short h0[CIB], h1[CIB];
      const bool rowin = z >= 0 && z < D && yy >= 0 && yy < H;
#pragma unroll
      for (int c = 0; c < CIB; ++c) { v0[c] = 0.f; v1[c] = 0.f; h0[c] = 0; h1[c] = 0; }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int xq = xx + q;
        if (rowin && xq >= 0 && xq < W) {
          const int64_t off = (vbase + ((int64_t)z * H + yy) * W + xq) * Cin + ci0;
          if (X_BF16) {
#pragma unroll
            for (int c8 = 0; c8 < CIB / 8; ++c8) {
              const uint4 v = *reinterpret_cast<const uint4*>((const unsigned short*)xin + off + c8 * 8);
              const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                (q ? h1 : h0)[c8 * 8 + 2 * j] = (unsigned short)(wds[j] & 0xffffu);
                (q ? h1 : h0)[c8 * 8 + 2 * j + 1] = (unsigned short)(wds[j] >> 16);
              }
            }
          } else {
#pragma unroll
            for (int c4 = 0; c4 < CIB / 4; ++c4) {
              const float4 v = *reinterpret_cast<const float4*>((const float*)xin + off + c4 * 4);
              (q ? v1 : v0)[c4 * 4] = v.x; (q ? v1 : v0)[c4 * 4 + 1] = v.y; (q ? v1 : v0)[c4 * 4 + 2] = v.z; (q ? v1 : v0)[c4 * 4 + 3] = v.w;
            }
          }
        }
      }
      unsigned* dst = reinterpret_cast<unsigned*>(xs) + ((hz * HY + hy) * WHXP) / 2 + p;
#pragma unroll
      for (int c = 0; c < CIB; ++c)
        dst[(c * PX) / 2] = X_BF16 ? ((unsigned)h0[c] | ((unsigned)h1[c] << 16)) : pack_bf16x2(v0[c], v1[c]);
    }
    // ---- d_y tile: [co][row][32], 16 voxel pairs per row; couts beyond Cout are zero
    for (int idx = tid; idx < ROWS * 16; idx += NTHR) {
      const int p = idx & 15, rr = idx >> 4;
      const int z = z0 + rr / TY, yy = y0 + rr % TY, xx = x0 + 2 * p;
      unsigned short h0[NCO], h1[NCO];
#pragma unroll
      for (int c = 0; c < NCO; ++c) { h0[c] = 0; h1[c] = 0; }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (z < D && yy < H && xx + q < W) {
          const int64_t off = (vbase + ((int64_t)z * H + yy) * W + xx + q) * Cout + co0;
#pragma unroll
          for (int c8 = 0; c8 < NCO / 8; ++c8) {
            if (co0 + c8 * 8 < Cout) {
              const uint4 v = *reinterpret_cast<const uint4*>(dy + off + c8 * 8);
              const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                (q ? h1 : h0)[c8 * 8 + 2 * j] = (unsigned short)(wds[j] & 0xffffu);
                (q ? h1 : h0)[c8 * 8 + 2 * j + 1] = (unsigned short)(wds[j] >> 16);
              }
            }
          }
        }
      }
      unsigned* dst = reinterpret_cast<unsigned*>(dys) + (rr * WTX) / 2 + p;
#pragma unroll
      for (int c = 0; c < NCO; ++c) dst[(c * PD) / 2] = (unsigned)h0[c] | ((unsigned)h1[c] << 16);
    }
    __syncthreads();
    // ---- k-steps: this wave's rows of the tile
    for (int rr = wave; rr < ROWS; rr += 4) {
      const int roff = ((rr / TY) * HY + (rr % TY)) * WHXP;
      uint4 bq[NTB];
#pragma unroll
      for (int n = 0; n < NTB; ++n)
        bq[n] = *reinterpret_cast<const uint4*>(dys + (n * 16 + li) * PD + rr * WTX + 8 * lk);
      if (do_bias) {
#pragma unroll
        for (int n = 0; n < NTB; ++n)
          accb[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bq[n]), accb[n], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned short* pa = xs + aoff[u] + roff;
        const unsigned p3 = *reinterpret_cast<const unsigned*>(pa + 6);
        const uint4 q = *reinterpret_cast<const uint4*>(pa + 8);
        const unsigned r0 = *reinterpret_cast<const unsigned*>(pa + 16);
        const unsigned a0 = __builtin_amdgcn_alignbit(q.x, p3, 16), a1 = __builtin_amdgcn_alignbit(q.y, q.x, 16),
                       a2 = __builtin_amdgcn_alignbit(q.z, q.y, 16), a3 = __builtin_amdgcn_alignbit(q.w, q.z, 16),
                       a4 = __builtin_amdgcn_alignbit(r0, q.w, 16);
        const bf16x8 f0 = __builtin_bit_cast(bf16x8, make_uint4(a0, a1, a2, a3));
        const bf16x8 f1 = __builtin_bit_cast(bf16x8, q);
        const bf16x8 f2 = __builtin_bit_cast(bf16x8, make_uint4(a1, a2, a3, a4));
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
          const bf16x8 bb = __builtin_bit_cast(bf16x8, bq[n]);
          acc[u][0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, bb, acc[u][0][n], 0, 0, 0);
          acc[u][1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, bb, acc[u][1][n], 0, 0, 0);
          acc[u][2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2, bb, acc[u][2][n], 0, 0, 0);
        }
      }
    }
  }
  // ---- sum the 4 waves through LDS (fixed order), one partial per workgroup
  float* red = reinterpret_cast<float*>(lds);
  for (int w4 = 0; w4 < 4; ++w4) {
    __syncthreads();
    if (wave == w4) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int n = 0; n < NTB; ++n) {
            float4* slot = reinterpret_cast<float4*>(red + ((u * 3 + d) * NTB + n) * 256) + lane;
            float4 v = make_float4(acc[u][d][n][0], acc[u][d][n][1], acc[u][d][n][2], acc[u][d][n][3]);
            if (w4 > 0) { const float4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *slot = v;
          }
#pragma unroll
      for (int n = 0; n < NTB; ++n) {
        float4* slot = reinterpret_cast<float4*>(red + (U * 3 * NTB + n) * 256) + lane;
        float4 v = make_float4(accb[n][0], accb[n][1], accb[n][2], accb[n][3]);
        if (w4 > 0) { const float4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *slot = v;
      }
    }
  }
  __syncthreads();
  float* out = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * RED_FL;
  for (int i = tid; i < RED_FL; i += NTHR) out[i] = red[i];
}

// stage 1 of the partial reduction: red[j] = sum_g part[g][j] over the gx workgroup partials, j over row_fl floats (a
// multiple of 256).  256 threads = (256 / RG) column QUADS x RG row lanes: a thread owns 4 consecutive columns (16-byte loads:
// a wave reads 1 KB of one partial row per instruction) and every RG-th row, eight rows in flight (fenced: one load -> wait
// -> add per row is a chain of dependent round trips), fp64 accumulators, the RG row lanes summed through LDS in fixed
// order: deterministic.  RG follows the number of partial rows (colsum_rg: 512 rows -> 16 lanes of 32 rows, 16 rows -> one
// lane): the first form (64 columns x 16 row lanes of scalar loads, whatever gx) ran the 390 MB of one step's partials at
// 2.9 TB/s, and layers with few partial rows used one load per thread.
__host__ __device__ inline int colsum_rg(int gx) { return gx >= 512 ? 16 : (gx >= 256 ? 8 : (gx >= 128 ? 4 : (gx >= 64 ? 2 : 1))); }
__host__ __device__ inline int colsum_blocks(int64_t row_fl, int gx) { return (int)cdiv64(row_fl / 4, 256 / colsum_rg(gx)); }
__device__ __forceinline__ void wgrad_bf16_colsum_body(const float* __restrict__ part, float* __restrict__ red, int gx,
                                                       int64_t row_fl, int blk, double (*sm)[4]) {
  const int RG = colsum_rg(gx), QPB = 256 / RG;
  const int q = threadIdx.x % QPB, rg = threadIdx.x / QPB;
  const int64_t j = ((int64_t)blk * QPB + q) * 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (j < row_fl) {
    const float* base = part + j;
    int g = rg;
    for (; g + 7 * RG < gx; g += 8 * RG) {
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = *reinterpret_cast<const float4*>(base + (int64_t)(g + RG * u) * row_fl);
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(r[u].x), "+v"(r[u].y), "+v"(r[u].z), "+v"(r[u].w));
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 += (double)r[u].x; a1 += (double)r[u].y; a2 += (double)r[u].z; a3 += (double)r[u].w; }
    }
    {
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = *reinterpret_cast<const float4*>(base + (int64_t)(g + RG * u < gx ? g + RG * u : 0) * row_fl);   // row 0 always exists
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(r[u].x), "+v"(r[u].y), "+v"(r[u].z), "+v"(r[u].w));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = g + RG * u < gx;
        a0 += in ? (double)r[u].x : 0.0; a1 += in ? (double)r[u].y : 0.0; a2 += in ? (double)r[u].z : 0.0; a3 += in ? (double)r[u].w : 0.0;
      }
    }
  }
  if (RG > 1) {
    sm[threadIdx.x][0] = a0; sm[threadIdx.x][1] = a1; sm[threadIdx.x][2] = a2; sm[threadIdx.x][3] = a3;
    __syncthreads();
    if (rg == 0) {
      for (int k = 1; k < RG; ++k) { a0 += sm[k * QPB + q][0]; a1 += sm[k * QPB + q][1]; a2 += sm[k * QPB + q][2]; a3 += sm[k * QPB + q][3]; }
    }
  }
  if (rg == 0 && j < row_fl) *reinterpret_cast<float4*>(red + j) = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
}
__global__ __launch_bounds__(256) void wgrad_bf16_colsum_kernel(const float* __restrict__ part, float* __restrict__ red,
                                                                int gx, int64_t row_fl) {
  __shared__ double sm[256][4];
  wgrad_bf16_colsum_body(part, red, gx, row_fl, blockIdx.x, sm);
}

// stage 2: d_w[co][ci][tap] / d_bias[co] gathered out of the reduced fragment-layout buffer (call with gx = 1).
__device__ __forceinline__ void wgrad_bf16_reduce_body(const float* __restrict__ part, float* __restrict__ dw,
                                                       float* __restrict__ dbias, int Cin, int Cout, int CIB, int U, int NTB,
                                                       int gx, int gy, int n_coblk, int blk, int layout = 0) {
  const int nW = Cout * Cin * 27;
  const int i = blk * 256 + threadIdx.x;
  if (i >= nW + Cout) return;
  if (layout == 1) {
    // N-packed tiles of conv_x3_wgrad_kernel (Cout <= 8): column n = q * 8 + co multiplies d_y shifted by q voxels in x, row
    // tile t in {0,1} is x shifted by t: (t, q) = (0,1), (0,0), (1,0) are the taps dx = 0, 1, 2; (1,1) is not a tap
    const int RED_FL = (U * 2 + 1) * 256;
    int co, slot, m, q = 0;
    if (i < nW) {
      const int tap = i % 27, ci = (i / 27) % Cin;
      co = i / (27 * Cin);
      const int combo = tap / 3, dx = tap % 3;
      const int gl = CIB == 8 ? combo / 2 : combo / 4;
      m = CIB == 8 ? (combo % 2) * 8 + ci : (combo % 4) * 4 + ci;
      slot = gl * 2 + (dx == 2 ? 1 : 0);
      q = dx == 0 ? 1 : 0;
    } else {
      co = i - nW; slot = U * 2; m = 0;
    }
    const int n = q * 8 + co, lane = (m / 4) * 16 + n, reg = m % 4;
    const size_t off = (size_t)slot * 256 + lane * 4 + reg;
    double a = 0.0;
    for (int g = 0; g < gx; ++g) a += (double)part[(size_t)g * RED_FL + off];
    if (i < nW) dw[i] = (float)a;
    else if (dbias) dbias[co] = (float)a;
    return;
  }
  if (layout == 2) {
    // conv_wgrad_tr_kernel (conv3d_wtr.hip): CIB = NQ quads per channel block, U = MT = 3 NF real M tiles (+ 1 bias slot),
    // NTB = NT; M tile mt = 3 f + dx, family f holds the chunks q = 4 f .. 4 f + 3 of the list q = (dz, dy) * NQ + quad,
    // row = 4 (q & 3) + channel % 4
    const int NQ = CIB, MT = U, NT = NTB;
    const int RED_FL = (MT + 1) * NT * 256;
    int co, by_ci = 0, mt, c = 0, e = 0;
    if (i < nW) {
      const int tap = i % 27, ci = (i / 27) % Cin;
      co = i / (27 * Cin);
      by_ci = ci / (4 * NQ);
      const int q = (tap / 3) * NQ + (ci % (4 * NQ)) / 4;
      mt = (q >> 2) * 3 + tap % 3; c = q & 3; e = ci & 3;
    } else {
      co = i - nW; mt = MT;
    }
    const int cob = co / (16 * NT), nt = (co / 16) % NT, j = co % 16;
    const size_t off = (size_t)(by_ci * n_coblk + cob) * RED_FL + (size_t)(mt * NT + nt) * 256 + (c * 16 + j) * 4 + e;
    double a = 0.0;
    for (int g = 0; g < gx; ++g) a += (double)part[(size_t)g * gy * RED_FL + off];
    if (i < nW) dw[i] = (float)a;
    else if (dbias) dbias[co] = (float)a;
    return;
  }
  const int SLOTS = U * 3 + 1, RED_FL = SLOTS * NTB * 256;
  const int GPB = CIB == 16 ? 9 : (CIB == 8 ? 5 : 3);
  int co, slot, m, chunk;
  if (i < nW) {
    const int tap = i % 27, ci = (i / 27) % Cin;
    co = i / (27 * Cin);
    const int combo = tap / 3, dx = tap % 3;
    int gl;
    if (CIB == 16) { gl = combo; m = ci % 16; }
    else if (CIB == 8) { gl = combo / 2; m = (combo % 2) * 8 + ci; }
    else { gl = combo / 4; m = (combo % 4) * 4 + ci; }
    const int gg = (CIB == 16 ? (ci / 16) * GPB : 0) + gl;
    chunk = gg / U;
    slot = (gg % U) * 3 + dx;
  } else {
    co = i - nW; chunk = 0; slot = U * 3; m = 0;
  }
  const int nbg = co / 16, yb = nbg / NTB, nb = nbg % NTB, n = co % 16;
  const int lane = (m / 4) * 16 + n, reg = m % 4;
  const size_t off = (size_t)(chunk * n_coblk + yb) * RED_FL + (size_t)(slot * NTB + nb) * 256 + lane * 4 + reg;
  double a = 0.0;
  for (int g = 0; g < gx; ++g) a += (double)part[(size_t)g * gy * RED_FL + off];
  if (i < nW) dw[i] = (float)a;
  else if (dbias) dbias[co] = (float)a;
}
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int Cin, int Cout, int CIB, int U,
                                                                int NTB, int gx, int gy, int n_coblk, int layout) {
  wgrad_bf16_reduce_body(part, dw, dbias, Cin, Cout, CIB, U, NTB, gx, gy, n_coblk, blockIdx.x, layout);
}

// Deferred form (modet_conv3d_bf16_bwd_weight_defer + modet_conv3d_wgrad_defer_flush): both stages of every queued
// weight gradient in two launches instead of two per layer; job tables by value in the kernel arguments.
constexpr int BRED_MAX_JOBS = 24;
struct BRedTable { BRedJob job[BRED_MAX_JOBS]; int first[BRED_MAX_JOBS + 1]; int n; };
__global__ __launch_bounds__(256) void wgrad_bf16_colsum_many_kernel(const BRedTable t) {
  __shared__ double sm[256][4];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  const BRedJob& J = t.job[j];
  wgrad_bf16_colsum_body(J.part, J.red, J.gx, J.row_fl, blockIdx.x - t.first[j], sm);
}
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_many_kernel(const BRedTable t) {
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  const BRedJob& J = t.job[j];
  wgrad_bf16_reduce_body(J.red, J.dw, J.dbias, J.Cin, J.Cout, J.cib, J.u, J.ntb, 1, J.gy, J.n_coblk, blockIdx.x - t.first[j], J.layout);
}

struct WgBf16Plan { int cib, u, ntb, tz, ty, n_chunk, n_coblk, gy, gx, ntiles, tiles_x, tiles_y, tiles_z, red_fl; };
inline WgBf16Plan plan_wgrad_bf16(int B, int D, int H, int W, int Cin, int Cout) {
  WgBf16Plan p;
  if (Cin <= 4) { p.cib = 4; p.u = 3; p.ntb = 1; p.tz = 4; p.ty = 8; }
  else if (Cin <= 8) { p.cib = 8; p.u = 5; p.ntb = 1; p.tz = 4; p.ty = 8; }
  else if (Cout <= 16) { p.cib = 16; p.u = 9; p.ntb = 1; p.tz = 2; p.ty = 8; }
  else { p.cib = 16; p.u = 3; p.ntb = 2; p.tz = 2; p.ty = 4; }
  const int gpb = p.cib == 16 ? 9 : (p.cib == 8 ? 5 : 3);
  const int nblk = p.cib == 16 ? cdiv(Cin, 16) : 1;
  p.n_chunk = nblk * gpb / p.u;
  p.n_coblk = cdiv(Cout, p.ntb * 16);
  p.gy = p.n_chunk * p.n_coblk;
  p.tiles_x = cdiv(W, WTX); p.tiles_y = cdiv(H, p.ty); p.tiles_z = cdiv(D, p.tz);
  p.ntiles = B * p.tiles_x * p.tiles_y * p.tiles_z;
  int gx = (256 * 2) / p.gy;
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  p.gx = gx;
  p.red_fl = (p.u * 3 + 1) * p.ntb * 256;
  return p;
}

struct Bf16Plan { int tz, ty, ck, nt, nstage, cinp, coutp, ksteps; };
inline int round_up_i(int a, int b) { return (a + b - 1) / b * b; }
inline Bf16Plan plan_bf16(int64_t V, int Cin, int Cout) {
  Bf16Plan p;
  p.cinp = round_up_i(Cin, 8);
  p.ck = p.cinp % 32 == 0 ? 32 : (p.cinp % 16 == 0 ? 16 : 8);
  p.nstage = p.cinp / p.ck;
  p.nt = Cout <= 16 ? 1 : 2;
  p.coutp = round_up_i(Cout, p.nt * 16);
  p.ksteps = (27 * p.ck + 31) / 32;
  if (false && V >= 500000 && p.ck <= 16 && p.nt == 1) { p.tz = 8; p.ty = 8; }
  else if (V >= 60000) { p.tz = 4; p.ty = 8; }
  else { p.tz = 2; p.ty = 4; }
  return p;
}

inline size_t bf16_wpk_elems(int Cin, int Cout) {
  // generous: any plan pads Cin to a multiple of 8 and the k index to 32 per step (7 steps per 8 channels, 14 per 16,
  // 27 per 32: at most 28 k slots per channel), Cout to a multiple of 32
  return (size_t)28 * round_up_i(Cin, 32) * round_up_i(Cout, 32) + 1024;
}

template <bool IN_BF16, bool OUT_BF16, bool STATS>
int launch_bf16(modet_step_ctx* step, const void* x, const float* w, const float* bias, void* y, void* ws, float* stats, int B,
                int D, int H, int W, int Cin, int Cout, int mode, hipStream_t s) {
  const int64_t V = (int64_t)D * H * W;
  const Bf16Plan p = plan_bf16(V, Cin, Cout);
  unsigned short* wpk = (unsigned short*)ws;
  const int total = p.nstage * p.ksteps * p.coutp * 32;
  if (const unsigned short* pre = prepacked_bf16_or_record(step, PackBKey{w, Cin, Cout, p.coutp, p.ck, p.nstage, p.ksteps, mode, 1, 0}))
    wpk = const_cast<unsigned short*>(pre);
  else
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0, s, w, wpk,
                       Cin, Cout, p.coutp, p.ck, p.nstage, p.ksteps, mode, 1);
  const int tiles_x = cdiv(W, TX), tiles_y = cdiv(H, p.ty), tiles_z = cdiv(D, p.tz);
  const dim3 grid(tiles_x * tiles_y * tiles_z, p.coutp / (p.nt * 16), B);
  float* shift = nullptr;
  float* rows = nullptr;
  if (STATS) {
    shift = stats;
    rows = stats + (size_t)B * Cout;
    hipLaunchKernelGGL(conv_shift_bf16_kernel<IN_BF16>, dim3(cdiv(B * Cout, 4)), dim3(256), 0, s, x, w, bias, shift, B, D, H, W,
                       Cin, Cout);
  }
#define BF_LAUNCH(TZ_, TY_, CK_, NT_)                                                                                     \
  hipLaunchKernelGGL((conv3d_bf16_kernel<TZ_, TY_, CK_, NT_, IN_BF16, OUT_BF16, STATS>), grid, dim3(NTHR), 0, s, x,        \
                     (const uint4*)wpk, bias, y, rows, (const float*)shift, D, H, W, Cin, Cout, p.coutp, p.nstage, tiles_x, tiles_y, 0)
#define BF_CK(TZ_, TY_, NT_)                                  \
  do {                                                        \
    if (p.ck == 8) BF_LAUNCH(TZ_, TY_, 8, NT_);               \
    else if (p.ck == 16) BF_LAUNCH(TZ_, TY_, 16, NT_);        \
    else BF_LAUNCH(TZ_, TY_, 32, NT_);                        \
  } while (0)
  if (p.tz == 8) {                       // ck <= 16, nt == 1 by construction
    if (p.ck == 8) BF_LAUNCH(8, 8, 8, 1); else BF_LAUNCH(8, 8, 16, 1);
  } else if (p.tz == 4) {
    if (p.nt == 1) BF_CK(4, 8, 1); else BF_CK(4, 8, 2);
  } else {
    if (p.nt == 1) BF_CK(2, 4, 1); else BF_CK(2, 4, 2);
  }
#undef BF_CK
#undef BF_LAUNCH
  return modet_launch_status();
}

// ---- fp32 emulation ("bf16x3", SP = 3): fp32 in, fp32 out.  Channel chunks of at most 16 (three LDS planes per tile).
inline Bf16Plan plan_split(int64_t V, int Cin, int Cout) {
  Bf16Plan p;
  p.cinp = round_up_i(Cin, 8);
  p.ck = p.cinp % 16 == 0 ? 16 : 8;
  p.nstage = p.cinp / p.ck;
  p.nt = Cout <= 16 ? 1 : 2;
  p.coutp = round_up_i(Cout, p.nt * 16);
  p.ksteps = (27 * p.ck + 31) / 32;
  if (V >= 60000) { p.tz = p.ck == 8 ? 4 : 2; p.ty = 8; }
  else { p.tz = 2; p.ty = 4; }
  return p;
}

template <bool STATS>
int launch_split(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats, int B,
                 int D, int H, int W, int Cin, int Cout, int mode, hipStream_t s) {
  const int64_t V = (int64_t)D * H * W;
  const Bf16Plan p = plan_split(V, Cin, Cout);
  unsigned short* wpk = (unsigned short*)ws;
  const int total = p.nstage * p.ksteps * p.coutp * 32;
  if (const unsigned short* pre = prepacked_bf16_or_record(step, PackBKey{w, Cin, Cout, p.coutp, p.ck, p.nstage, p.ksteps, mode, 3, 0}))
    wpk = const_cast<unsigned short*>(pre);
  else
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0, s, w, wpk,
                       Cin, Cout, p.coutp, p.ck, p.nstage, p.ksteps, mode, 3);
  const int tiles_x = cdiv(W, TX), tiles_y = cdiv(H, p.ty), tiles_z = cdiv(D, p.tz);
  const dim3 grid(tiles_x * tiles_y * tiles_z, p.coutp / (p.nt * 16), B);
  float* shift = nullptr;
  float* rows = nullptr;
  if (STATS) {
    shift = stats;
    rows = stats + (size_t)B * Cout;
    hipLaunchKernelGGL(conv_shift_bf16_kernel<false>, dim3(cdiv(B * Cout, 4)), dim3(256), 0, s, (const void*)x, w, bias, shift, B, D,
                       H, W, Cin, Cout);
  }
  const int wpiece = total / 8;                            // uint4 units between the hi / mid / lo weight arrays
#define SP_LAUNCH(TZ_, TY_, CK_, NT_)                                                                                     \
  hipLaunchKernelGGL((conv3d_bf16_kernel<TZ_, TY_, CK_, NT_, false, false, STATS, 3>), grid, dim3(NTHR), 0, s, (const void*)x, \
                     (const uint4*)wpk, bias, (void*)y, rows, (const float*)shift, D, H, W, Cin, Cout, p.coutp, p.nstage, tiles_x, \
                     tiles_y, wpiece)
  if (p.tz == 4) { if (p.nt == 1) SP_LAUNCH(4, 8, 8, 1); else SP_LAUNCH(4, 8, 8, 2); }
  else if (p.ty == 8) { if (p.nt == 1) SP_LAUNCH(2, 8, 16, 1); else SP_LAUNCH(2, 8, 16, 2); }
  else if (p.ck == 8) { if (p.nt == 1) SP_LAUNCH(2, 4, 8, 1); else SP_LAUNCH(2, 4, 8, 2); }
  else { if (p.nt == 1) SP_LAUNCH(2, 4, 16, 1); else SP_LAUNCH(2, 4, 16, 2); }
#undef SP_LAUNCH
  return modet_launch_status();
}

inline int bf16_tiles_per_sample(int D, int H, int W, int Cin, int Cout) {
  const Bf16Plan p = plan_bf16((int64_t)D * H * W, Cin, Cout);
  return cdiv(W, TX) * cdiv(H, p.ty) * cdiv(D, p.tz);
}

// MODET_CONV_X3 = 0 keeps the tiled kernels (A/B measurements); otherwise the z-marching kernel takes the few-channel layers
inline bool x3_on() {
  static const bool on = modet_tuning_env("MODET_CONV_X3") != '0';
  return on;
}

}  // namespace

// ---- internal interface for conv3d.hip (C++ linkage, not part of the ABI): the fp32 entry points route eligible shapes here
bool modetx_split_eligible(int Cin, int Cout) { return Cin > 1 && Cin % 4 == 0 && Cout % 4 == 0; }
size_t modetx_split_ws_bytes(int Cin, int Cout) {
  const int m = Cin > Cout ? Cin : Cout;
  return 3 * bf16_wpk_elems(m, m) * sizeof(unsigned short);
}
size_t modetx_split_stats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const Bf16Plan p = plan_split((int64_t)D * H * W, Cin, Cout);
  const size_t tiles = (size_t)cdiv(W, TX) * cdiv(H, p.ty) * cdiv(D, p.tz);
  return ((size_t)B * Cout + (size_t)B * tiles * Cout * 2) * sizeof(float);
}
int modetx_split_conv(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats,
                      int B, int D, int H, int W, int Cin, int Cout, int mode, hipStream_t s) {
  return stats ? launch_split<true>(step, x, w, bias, y, ws, stats, B, D, H, W, Cin, Cout, mode, s)
               : launch_split<false>(step, x, w, bias, y, ws, nullptr, B, D, H, W, Cin, Cout, mode, s);
}

// ---- conv3d_x3.hip's weight-gradient kernel writes this file's partial layout (fragment-major tiles, slot = group * 3 + dx,
// bias slot last): its two reduction stages run here, immediately or queued in the caller's step context
int modetx_wgrad_partials_reduce(modet_step_ctx* defer, const float* part, float* red, float* dw, float* db, int gx, int Cin,
                                 int Cout, int cib, int u, int layout, hipStream_t s) {
  const int red_fl = (u * (layout == 1 ? 2 : 3) + 1) * 256;     // ntb = 1, gy = 1, n_coblk = 1
  const int64_t row_fl = red_fl;
  if (defer) {
    std::lock_guard<std::mutex> lk(defer->mu);
    defer->brjobs.push_back(BRedJob{part, red, dw, db, row_fl, gx, Cin, Cout, cib, u, 1, 1, 1, layout});
    return modet_launch_status();
  }
  hipLaunchKernelGGL(wgrad_bf16_colsum_kernel, dim3(colsum_blocks(row_fl, gx)), dim3(256), 0, s, part, red, gx, row_fl);
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(cdiv(Cout * Cin * 27 + Cout, 256)), dim3(256), 0, s, (const float*)red, dw, db,
                     Cin, Cout, cib, u, 1, 1, 1, 1, layout);
  return modet_launch_status();
}

// ---- conv3d_wtr.hip's partial layout (layout 2): gy = (channel blocks) x (cout blocks) partial sets per workgroup row
int modetx_wgrad_partials_reduce2(modet_step_ctx* defer, const float* part, float* red, float* dw, float* db, int gx, int gy,
                                  int Cin, int Cout, int nq, int mt, int nt, int n_coblk, hipStream_t s) {
  const int64_t row_fl = (int64_t)gy * (mt + 1) * nt * 256;
  if (defer) {
    std::lock_guard<std::mutex> lk(defer->mu);
    defer->brjobs.push_back(BRedJob{part, red, dw, db, row_fl, gx, Cin, Cout, nq, mt, nt, gy, n_coblk, 2});
    return modet_launch_status();
  }
  hipLaunchKernelGGL(wgrad_bf16_colsum_kernel, dim3(colsum_blocks(row_fl, gx)), dim3(256), 0, s, part, red, gx, row_fl);
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(cdiv(Cout * Cin * 27 + Cout, 256)), dim3(256), 0, s, (const float*)red, dw, db,
                     Cin, Cout, nq, mt, nt, 1, gy, n_coblk, 2);
  return modet_launch_status();
}

// ---- bf16 side of modet_conv3d_wgrad_defer_flush (conv3d.hip calls it)
void modetx_bf16_defer_flush(modet_step_ctx* c, hipStream_t stream) {
  std::vector<BRedJob> jobs;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    jobs.swap(c->brjobs);
  }
  for (size_t i0 = 0; i0 < jobs.size(); i0 += BRED_MAX_JOBS) {
    BRedTable t;
    const int n = (int)(jobs.size() - i0 < (size_t)BRED_MAX_JOBS ? jobs.size() - i0 : (size_t)BRED_MAX_JOBS);
    t.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) { t.job[i] = jobs[i0 + i]; t.first[i] = total; total += colsum_blocks(t.job[i].row_fl, t.job[i].gx); }
    for (int i = n; i <= BRED_MAX_JOBS; ++i) t.first[i] = total;
    hipLaunchKernelGGL(wgrad_bf16_colsum_many_kernel, dim3(total), dim3(256), 0, stream, t);
    total = 0;
    for (int i = 0; i < n; ++i) { t.first[i] = total; total += cdiv(t.job[i].Cout * t.job[i].Cin * 27 + t.job[i].Cout, 256); }
    for (int i = n; i <= BRED_MAX_JOBS; ++i) t.first[i] = total;
    hipLaunchKernelGGL(wgrad_bf16_reduce_many_kernel, dim3(total), dim3(256), 0, stream, t);
  }
}

// ---- 16-bit side of modet_conv3d_prepack_* (conv3d.hip owns the entry points; these jobs follow the fp32 jobs in the arena)
void modetx_x3_prepack_begin(modet_step_ctx* c, hipStream_t stream);          // conv3d_x3.hip: its jobs have layout >= 2
bool modetx_x3_bf16_eligible(int B, int D, int H, int W, int Cin, int Cout, int x_bf16);
int modetx_x3_bf16_rows_per_sample(int B, int D, int H, int W, int Cin, int Cout);
int modetx_x3_bf16_conv(modet_step_ctx* step, const void* x, int x_bf16, const float* w, const float* bias, void* y, int y_bf16,
                        void* ws, float* stats, int B, int D, int H, int W, int Cin, int Cout, int mode, hipStream_t s);
bool modetx_x3_bf16_wgrad_eligible(int B, int D, int H, int W, int Cin, int Cout, int x_bf16);
size_t modetx_x3_bf16_wgrad_ws_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modetx_x3_bf16_wgrad(modet_step_ctx* defer, const void* x, int x_bf16, const void* dy, float* dw, float* db, void* ws, int B,
                         int D, int H, int W, int Cin, int Cout, hipStream_t s);
size_t modetx_bf16_prepack_bytes(modet_step_ctx* c) {
  std::lock_guard<std::mutex> lk(c->mu);
  size_t n = 0;
  for (const PackBKey& k : c->bjobs) n += packb_elems(k);
  return n * sizeof(unsigned short);
}
void modetx_bf16_prepack_begin(modet_step_ctx* c, void* arena, hipStream_t stream) {
  std::vector<PackBKey> jobs;
  std::vector<size_t> off;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->bjobs.empty()) return;
    c->boff.assign(c->bjobs.size(), 0);
    size_t n = 0;
    for (size_t i = 0; i < c->bjobs.size(); ++i) { c->boff[i] = n; n += packb_elems(c->bjobs[i]); }
    c->barena = (unsigned short*)arena;
    for (size_t i = 0; i < c->bjobs.size(); ++i)
      if (c->bjobs[i].layout < 2) { jobs.push_back(c->bjobs[i]); off.push_back(c->boff[i]); }
  }
  for (size_t i0 = 0; i0 < jobs.size(); i0 += PACKB_MAX_JOBS) {
    PackBTable t;
    const int n = (int)(jobs.size() - i0 < (size_t)PACKB_MAX_JOBS ? jobs.size() - i0 : (size_t)PACKB_MAX_JOBS);
    int most = 1;
    for (int i = 0; i < n; ++i) {
      const PackBKey& k = jobs[i0 + i];
      t.job[i] = PackBJob{k.w, (unsigned short*)arena + off[i0 + i], k.Cin, k.Cout, k.CoutP, k.CK, k.nstage, k.ksteps, k.mode, k.npiece};
      const int blocks = cdiv(k.nstage * k.ksteps * k.CoutP * 32, 256);
      most = blocks > most ? blocks : most;
    }
    t.n = n;
    hipLaunchKernelGGL(pack_weights_bf16_many_kernel, dim3(most > 64 ? 64 : most, n), dim3(256), 0, stream, t);
  }
  modetx_x3_prepack_begin(c, stream);
}

extern "C" {

int modet_conv3d_bf16_kernel_family(int B, int D, int H, int W, int Cin, int Cout, int pass, int x_bf16) {
  if (!x3_on()) return 1;
  if (pass == 0) return modetx_x3_bf16_eligible(B, D, H, W, Cin, Cout, x_bf16) ? 2 : 1;
  if (pass == 1) return modetx_x3_bf16_eligible(B, D, H, W, Cout, Cin, 1) ? 2 : 1;
  return modetx_x3_bf16_wgrad_eligible(B, D, H, W, Cin, Cout, x_bf16) ? 2 : 1;
}

size_t modet_conv3d_bf16_ws_bytes(int Cin, int Cout) {
  const int m = Cin > Cout ? Cin : Cout;
  return bf16_wpk_elems(m, m) * sizeof(unsigned short);
}

size_t modet_conv3d_bf16_stats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  if (Cout % 4 != 0 || Cout > 128) return 0;
  // [sample][Cout] shift header, one row [Cout][2] per (sample, output tile), and a tail of 64 rows per sample for the
  // first stage of modet_instnorm_lrelu_fwd_stats_bf16's reduction
  // the row count of the kernel with FEWER rows would do for either input type; fp32 and bf16 inputs share a plan anyway
  const int rows = (x3_on() && modetx_x3_bf16_eligible(B, D, H, W, Cin, Cout, Cin % 8 == 0)) ? modetx_x3_bf16_rows_per_sample(B, D, H, W, Cin, Cout)
                                                                                         : bf16_tiles_per_sample(D, H, W, Cin, Cout);
  return ((size_t)B * Cout + (size_t)B * (rows + 64) * Cout * 2) * sizeof(float);
}

int modet_conv3d_bf16_fwd(const void* x, int x_bf16, const float* w, const float* bias, void* y, void* ws, size_t ws_bytes,
                          float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                          modet_stream_t stream, modet_step_ctx_t* step) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(w); MODET_CHECK_PTR(y); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && B <= 65535);
  if (Cout % 8 != 0 || (x_bf16 ? Cin % 8 != 0 : Cin % 4 != 0)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < bf16_wpk_elems(Cin, Cout) * sizeof(unsigned short)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (stats && (stats_bytes < modet_conv3d_bf16_stats_bytes(B, D, H, W, Cin, Cout) || stats_bytes == 0)) return MODET_ERR_WORKSPACE;
  if (x3_on() && modetx_x3_bf16_eligible(B, D, H, W, Cin, Cout, x_bf16)) {
    if (stats) {
      if (x_bf16) hipLaunchKernelGGL(conv_shift_bf16_kernel<true>, dim3(cdiv(B * Cout, 4)), dim3(256), 0, s, x, w, bias, stats, B, D, H, W, Cin, Cout);
      else hipLaunchKernelGGL(conv_shift_bf16_kernel<false>, dim3(cdiv(B * Cout, 4)), dim3(256), 0, s, x, w, bias, stats, B, D, H, W, Cin, Cout);
    }
    return modetx_x3_bf16_conv(step, x, x_bf16, w, bias, y, 1, ws, stats, B, D, H, W, Cin, Cout, 0, s);
  }
  if (stats) {
    return x_bf16 ? launch_bf16<true, true, true>(step, x, w, bias, y, ws, stats, B, D, H, W, Cin, Cout, 0, s)
                  : launch_bf16<false, true, true>(step, x, w, bias, y, ws, stats, B, D, H, W, Cin, Cout, 0, s);
  }
  return x_bf16 ? launch_bf16<true, true, false>(step, x, w, bias, y, ws, nullptr, B, D, H, W, Cin, Cout, 0, s)
                : launch_bf16<false, true, false>(step, x, w, bias, y, ws, nullptr, B, D, H, W, Cin, Cout, 0, s);
}

int modet_conv3d_bf16_bwd_data(const void* d_y, const float* w, void* d_x, int dx_bf16, void* ws, size_t ws_bytes, int B, int D,
                               int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(w); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && B <= 65535);
  if (Cout % 8 != 0 || (dx_bf16 ? Cin % 8 != 0 : Cin % 4 != 0)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < bf16_wpk_elems(Cout, Cin) * sizeof(unsigned short)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // a convolution of d_y (Cout channels, bf16) producing Cin channels
  if (x3_on() && modetx_x3_bf16_eligible(B, D, H, W, Cout, Cin, 1))
    return modetx_x3_bf16_conv(step, d_y, 1, w, nullptr, d_x, dx_bf16, ws, nullptr, B, D, H, W, Cout, Cin, 1, s);
  return dx_bf16 ? launch_bf16<true, true, false>(step, d_y, w, nullptr, d_x, ws, nullptr, B, D, H, W, Cout, Cin, 1, s)
                 : launch_bf16<true, false, false>(step, d_y, w, nullptr, d_x, ws, nullptr, B, D, H, W, Cout, Cin, 1, s);
}

size_t modet_conv3d_bf16_bwd_weight_ws_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const WgBf16Plan p = plan_wgrad_bf16(B, D, H, W, Cin, Cout);
  size_t need = ((size_t)p.gx + 1) * p.gy * p.red_fl * sizeof(float);        // workgroup partials + their column sums
  if (x3_on() && modetx_x3_bf16_wgrad_eligible(B, D, H, W, Cin, Cout, 0)) {
    const size_t x3 = modetx_x3_bf16_wgrad_ws_bytes(B, D, H, W, Cin, Cout);
    need = x3 > need ? x3 : need;
  }
  return need;
}

static int bf16_bwd_weight_impl(const void* x, int x_bf16, const void* d_y, float* d_w, float* d_bias, void* ws,
                                size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                modet_step_ctx* defer);

int modet_conv3d_bf16_bwd_weight(const void* x, int x_bf16, const void* d_y, float* d_w, float* d_bias, void* ws,
                                 size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream) {
  return bf16_bwd_weight_impl(x, x_bf16, d_y, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream, nullptr);
}
int modet_conv3d_bf16_bwd_weight_defer(const void* x, int x_bf16, const void* d_y, float* d_w, float* d_bias, void* ws,
                                       size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                       modet_step_ctx_t* step) {
  MODET_CHECK_PTR(step);
  return bf16_bwd_weight_impl(x, x_bf16, d_y, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream, step);
}

static int bf16_bwd_weight_impl(const void* x, int x_bf16, const void* d_y, float* d_w, float* d_bias, void* ws,
                                size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                modet_step_ctx* defer) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(d_w); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (Cout % 8 != 0 || (x_bf16 ? Cin % 8 != 0 : Cin % 4 != 0)) return MODET_ERR_UNSUPPORTED;
  if (Cin > 8 && Cin % 16 != 0) return MODET_ERR_UNSUPPORTED;            // channel blocks of 16 beyond Cin = 8
  if (Cin == 4 && x_bf16) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_conv3d_bf16_bwd_weight_ws_bytes(B, D, H, W, Cin, Cout)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // the few-channel full-resolution layers: z-marching kernel (conv3d_x3.hip), one bf16 piece per operand
  if (x3_on() && modetx_x3_bf16_wgrad_eligible(B, D, H, W, Cin, Cout, x_bf16))
    return modetx_x3_bf16_wgrad(defer, x, x_bf16, d_y, d_w, d_bias, ws, B, D, H, W, Cin, Cout, s);
  const WgBf16Plan p = plan_wgrad_bf16(B, D, H, W, Cin, Cout);
  const dim3 grid(p.gx, p.gy);
#define WG_BF(CIB_, U_, NTB_, TZ_, TY_)                                                                                     \
  do {                                                                                                                     \
    if (x_bf16) hipLaunchKernelGGL((conv3d_bf16_wgrad_kernel<CIB_, U_, NTB_, TZ_, TY_, true>), grid, dim3(NTHR), 0, s, x,  \
                                   (const unsigned short*)d_y, (float*)ws, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, \
                                   p.ntiles, p.n_coblk);                                                                    \
    else hipLaunchKernelGGL((conv3d_bf16_wgrad_kernel<CIB_, U_, NTB_, TZ_, TY_, false>), grid, dim3(NTHR), 0, s, x,        \
                            (const unsigned short*)d_y, (float*)ws, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z,    \
                            p.ntiles, p.n_coblk);                                                                           \
  } while (0)
  if (p.cib == 4) hipLaunchKernelGGL((conv3d_bf16_wgrad_kernel<4, 3, 1, 4, 8, false>), grid, dim3(NTHR), 0, s, x,
                                     (const unsigned short*)d_y, (float*)ws, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z,
                                     p.ntiles, p.n_coblk);
  else if (p.cib == 8) WG_BF(8, 5, 1, 4, 8);
  else if (p.u == 9) WG_BF(16, 9, 1, 2, 8);
  else WG_BF(16, 3, 2, 2, 4);
#undef WG_BF
  const int nout = Cout * Cin * 27 + Cout;
  const int64_t row_fl = (int64_t)p.gy * p.red_fl;
  float* red = (float*)ws + (size_t)p.gx * row_fl;
  if (defer) {
    std::lock_guard<std::mutex> lk(defer->mu);
    defer->brjobs.push_back(BRedJob{(const float*)ws, red, d_w, d_bias, row_fl, p.gx, Cin, Cout, p.cib, p.u, p.ntb, p.gy, p.n_coblk, 0});
    return modet_launch_status();
  }
  hipLaunchKernelGGL(wgrad_bf16_colsum_kernel, dim3(colsum_blocks(row_fl, p.gx)), dim3(256), 0, s, (const float*)ws, red, p.gx, row_fl);
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, s, (const float*)red, d_w, d_bias, Cin, Cout,
                     p.cib, p.u, p.ntb, 1, p.gy, p.n_coblk, 0);
  return modet_launch_status();
}

}  // extern "C"
