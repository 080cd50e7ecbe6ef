// 3x3x3 / stride 1 / zero-pad 1 convolution, forward and data gradient, for the MID AND COARSE pyramid levels and the CWM
// layers (volumes of 1 200 .. 600 k voxels, 6 .. 128 channels) in fp32 accuracy on gfx950's bf16 matrix pipe ("bf16x3", see
// conv3d_x3.hip), with the K index packed in channel QUADS.
//   reference call sites: nn.Conv3d in ConvInsBlock / CWM, ModeT/models.py:135-151, :249-256 (arithmetic lives in ATen/MIOpen
//   there).
//
// Why another conv kernel.  Before it these ~35 launches of a train step ran on three kernels, each good at something else:
// the tiled bf16x3 kernel (conv3d_bf16.hip, SP = 3: 4x8x16-voxel tiles, one workgroup per tile -- at level 3 a 40-wide
// volume wastes 17 % of every 16-wide x tile and 720 tiles over 512 slots leave the second round half empty: 100 TFLOP/s),
// the exact-f32 MFMA kernel (conv3d.hip: every launch with a lazily normalised input, every odd channel count: 38 TFLOP/s)
// and the direct kernel (level 5: operands straight from L2, 50 TFLOP/s).  This one covers all of them but level 5 (2 400 voxels,
// where the direct kernel stays faster):
//   * tile = 2 x 8 x 8 voxels (x tiles of 8 fit W = 40 / 80 exactly), staged once per 16-channel block into the tensor's own
//     layout [halo'd voxel][16 ch] bf16 x three pieces (the staging code of conv3d_wtr.hip, plus the lazily applied
//     InstanceNorm + LeakyReLU of modet_conv3d_fwd_normin);
//   * k = (tap, channel quad): a k-step is 8 chunks of the list q = tap * NQ + quad, so 12 / 24 / 48 channels (NQ = 3) and
//     6 (NQ = 2) pack without padding; a lane's 8 k values are two 8-byte LDS reads at offsets taken from a 27 NQ-entry
//     table in LDS (a tap is an address offset);
//   * the MFMA takes the WEIGHTS as A (16 couts) and 16 VOXELS (2 rows x 8 x) as B, so a lane ends with 4 consecutive
//     couts of one voxel: bias, statistics and a 16-byte store without a transpose; weights come pre-split and
//     pre-arranged in fragment order from L2 (16 bytes per lane and (k-step, piece, cout tile), shared by the four waves
//     through L1) through a ring of 3-4 k-steps of lookahead; the next channel block's tile is loaded during the k-steps;
//   * workgroup = (tile, block of cout tiles), wave = 2 voxel tiles x 1 | 2 cout tiles (template <WC, CT>, chosen by
//     q_plan from a measured sweep); the channel blocks (stages) accumulate in registers; fused InstanceNorm statistics
//     as shifted sums, one row per (sample, tile).
// dgrad = the same kernel on flipped / transposed weights (mode 1).
#include "common.h"
#include "step_ctx.h"
#include <mutex>
#include <vector>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;

constexpr int NTHR = 256;
constexpr int TZ = 2, TY = 8, TX = 8, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int HVOX = HZ * HY * HX, VOX = TZ * TY * TX;
constexpr int ROWB = 32;                               // bytes of one voxel slot of a 16-channel bf16 image
constexpr int PITCH = HX * ROWB + 16;                  // bytes per halo'd x row: 16 mod 32, so the two rows a 32-lane half reads
                                                       // (8-byte accesses at quads q and q + 1) never meet on a bank
constexpr int XPL = HZ * HY * PITCH;                   // one piece of the x tile
constexpr unsigned Q_OOB = 0x80000000u;                // tensors are < 2 GiB (checked on the host): this offset reads 0

__device__ __forceinline__ BufRsrc q_rsrc(const void* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void split3_pk(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  auto pk = [](float u, float v) -> unsigned {
    const bf16x2 t = __builtin_convertvector((f32x2){u, v}, bf16x2);
    return __builtin_bit_cast(unsigned, t);
  };
  hi = pk(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = pk(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
  lo = pk(sa, sb);
}
// ---- two f16 pieces (round 5, see conv3d_x3.hip split2_h): x = hi + lo, hi = f16(x), lo = f16(x - hi); a product is the three
// piece products hi*hi + hi*lo + lo*hi.  f16 has the mantissa (x is carried to 2^-22 |x|) but not the range: weights are scaled
// by 2^8, activations by 2^4 (LeakyReLU(InstanceNorm(.)) <= sqrt(V), pooled / upsampled copies of it, flow fields: < 4 094),
// gradients by the power of two their producer's maximum gives (QArgs::amax), the accumulator is scaled back -- all exact.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr float Q_F16_WSCALE = 256.f, Q_F16_XSCALE = 16.f;
#ifndef Q_F16
#define Q_F16 1                                        // 0: three bf16 pieces everywhere (A/B builds)
#endif
__device__ __forceinline__ unsigned q_pk_f16(float u, float v) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const h16x2 t = __builtin_convertvector((f32x2){u, v}, h16x2);
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void split2h_pk(float a, float b, unsigned& hi, unsigned& lo) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
  hi = q_pk_f16(a, b);
  const h16x2 t = __builtin_bit_cast(h16x2, hi);
  lo = q_pk_f16(a - (float)t[0], b - (float)t[1]);
}
template <bool F16>
__device__ __forceinline__ f32x4 q_mma(bf16x8 a, bf16x8 b, f32x4 c) {       // (F16: the 16-byte fragments hold f16)
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned short q_bf16(float a) {
  const __bf16 x = (__bf16)a;
  return __builtin_bit_cast(unsigned short, x);
}

// ------------------------------------------------------------------------------------------------ weight packing
// wpk[(((stage * KS + ks) * 3 + piece) * CT + ct) * 64 + lane][8]: the MFMA A fragment of cout tile ct: row co = ct * 16 +
// (lane & 15), k = 8 (lane >> 4) + j = chunk 8 ks + 2 (lane >> 4) + (j >> 2), channel j & 3 of that chunk's quad.
//   chunk q -> tap = q / NQ, quad = q % NQ, channel ci = stage * 4 NQ + 4 quad + (j & 3); zero for q >= 27 NQ, ci >= Cin, co >= Cout.
//   mode 0 (forward): w[co][ci][tap]   w: (Cout, Cin, 27);   mode 1 (dgrad): w[ci][co][26 - tap]   w: (Co = ci, Ci = co, 27)
struct QPackJob { const float* w; unsigned short* wpk; int Cin, Cout, nq, nstage, ks, ct, mode, np; };   // np 3: bf16 pieces | 2: f16 pieces
__device__ __forceinline__ void q_pack_body(const QPackJob& J, int i0, int stride) {
  const int total = J.nstage * J.ks * J.ct * 64 * 8;   // elements of ONE piece
  for (int i = i0; i < total; i += stride) {
    const int j = i & 7, lane = (i >> 3) & 63;
    int t = i >> 9;
    const int ct = t % J.ct; t /= J.ct;
    const int ks = t % J.ks, stage = t / J.ks;
    const int q = 8 * ks + 2 * (lane >> 4) + (j >> 2);
    const int tap = q / J.nq, quad = q - tap * J.nq;
    const int ci = stage * 4 * J.nq + 4 * quad + (j & 3), co = ct * 16 + (lane & 15);
    float v = 0.f;
    if (q < 27 * J.nq && ci < J.Cin && co < J.Cout)
      v = J.mode == 0 ? J.w[((int64_t)co * J.Cin + ci) * 27 + tap] : J.w[((int64_t)ci * J.Cout + co) * 27 + 26 - tap];
    // piece p of (stage, ks) sits CT * 512 elements after piece p - 1
    const size_t base = ((size_t)((stage * J.ks + ks) * J.np) * J.ct + ct) * 512 + lane * 8 + j;
    if (J.np == 2) {
      const _Float16 hh = (_Float16)(v * Q_F16_WSCALE);
      const _Float16 ll = (_Float16)(v * Q_F16_WSCALE - (float)hh);
      J.wpk[base] = __builtin_bit_cast(unsigned short, hh);
      J.wpk[base + (size_t)J.ct * 512] = __builtin_bit_cast(unsigned short, ll);
      continue;
    }
    const unsigned short h = q_bf16(v);
    const float r1 = v - __uint_as_float((unsigned)h << 16);
    const unsigned short m = q_bf16(r1);
    const float r2 = r1 - __uint_as_float((unsigned)m << 16);
    J.wpk[base] = h;
    J.wpk[base + (size_t)J.ct * 512] = m;
    J.wpk[base + (size_t)2 * J.ct * 512] = q_bf16(r2);
  }
}
__global__ void q_pack_kernel(const QPackJob J) { q_pack_body(J, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }
constexpr int QPACK_MAX_JOBS = 48;
struct QPackTable { QPackJob job[QPACK_MAX_JOBS]; int n; };
__global__ void q_pack_many_kernel(const QPackTable t) {
  q_pack_body(t.job[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

struct QArgs {
  const float* x; const uint4* wpk; const float* bias; float* y;
  const float* in_mean; const float* in_rstd;          // NORM: the input is LeakyReLU((x - mean) * rstd), applied while staging
  float* stats_rows; const float* shift;               // STATS 1: rows [B][tiles][Cout][2] of sum(y - K), sum((y - K)^2); K = shift[b][co]
  const float* xraw; const float* bmean; const float* brstd;   // STATS 2 (a data gradient whose output is the gradient w.r.t.
                                                       // LeakyReLU(InstanceNorm(xraw))): rows of sum g, sum g * xhat, g = y * lrelu'(xhat)
                                                       // -- the first pass of that norm's backward (as conv3d_x3.hip STATS = 2)
  int B, D, H, W, Cin, Cout, tiles_x, tiles_y, tiles_z, nstage, ct_total;
  const float* amax;                                   // f16 pieces, x = a gradient: MODET_AMAX_SLOTS maxima of |x| (else null: x 2^4)
};

// Wave tiling <WC, CT>: the four waves form WC cout groups x (4 / WC) voxel groups; a wave owns CT cout tiles (cout block of
// the workgroup = WC * CT tiles) and VT = 2 WC voxel tiles.  Per k-step it reads VT x 3 pieces x 2 x 8 bytes of voxels from LDS
// and CT x 3 x 16 bytes of weights from L2 / L1 for VT * CT * 6 MFMAs; q_plan picks the tiling per shape (measured).
template <int NQ, int WC, int CT, bool NORM, int STATS, int NP = 3>
__global__ __launch_bounds__(NTHR, (2 * WC * CT >= 8) ? 2 : 3) void conv_q_kernel(const QArgs a) {
  static_assert(NP == 3 || NP == 2, "three bf16 pieces or two f16 pieces");
  constexpr int VT = 2 * WC;                           // voxel tiles per wave
  constexpr int CB = WC * CT;                          // cout tiles per workgroup
  constexpr int PF = CT == 1 ? 4 : (CT == 2 ? 3 : 2);   // k-steps of weight lookahead
  constexpr int KS = (27 * NQ + 7) / 8;                // k-steps per channel block
  constexpr int XS_BYTES = NP * XPL;
  constexpr int TAB = 27 * NQ + 8;                     // chunk -> LDS byte offset (tap, quad); the tail entries are dummies
  __shared__ __attribute__((aligned(16))) unsigned char lds[XS_BYTES + TAB * 4 + (STATS ? 4 * CT * 16 * 2 * 4 : 0)];
  int* tab = reinterpret_cast<int*>(lds + XS_BYTES);
  float* sred = reinterpret_cast<float*>(lds + XS_BYTES + TAB * 4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vj = lane & 15, kg = lane >> 4;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  int t = blockIdx.x;
  const int x0 = (t % a.tiles_x) * TX; t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * TY; t /= a.tiles_y;
  const int z0 = (t % a.tiles_z) * TZ;
  const int b = t / a.tiles_z;
  const int ctw = blockIdx.y * CB + (wave % WC) * CT;   // this wave's first cout tile
  const int vg = wave / WC;                            // its voxel group: voxel tiles vg * VT .. + VT - 1

  for (int i = tid; i < TAB; i += NTHR) {
    const int q = i < 27 * NQ ? i : 0;
    const int tap = q / NQ, quad = q - tap * NQ;
    tab[i] = ((tap / 9) * HY + (tap / 3) % 3) * PITCH + (tap % 3) * ROWB + quad * 8;
  }
  // this lane's voxel tiles: vt = vg * VT + v, voxel (z = vt >> 2, y = 2 (vt & 3) + (vj >> 3), x = vj & 7)
  int voff[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) {
    const int vt = vg * VT + v;
    voff[v] = ((vt >> 2) * HY + 2 * (vt & 3) + (vj >> 3)) * PITCH + (vj & 7) * ROWB;
  }

  f32x4 acc[VT][CT];
#pragma unroll
  for (int v = 0; v < VT; ++v)
#pragma unroll
    for (int n = 0; n < CT; ++n) acc[v][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const BufRsrc rx = q_rsrc(a.x, (unsigned)((int64_t)a.B * D * H * W * Cin * 4));
  // staging roles (as conv3d_wtr.hip): the halo'd tile is 40 rows (hz, hy) of IPR = 10 NQ 16-byte items; a pass covers RP rows
  constexpr int IPR = HX * NQ, RP = NQ >= 3 ? 5 : (NQ == 2 ? 10 : 20), NPASS = HZ * HY / RP;
  const int r_t = tid / IPR, i_t = tid - r_t * IPR;
  const bool x_act = r_t < RP;
  const int hx_t = i_t / NQ, qd_t = i_t - hx_t * NQ;
  const int hzo_t = RP == 20 ? r_t / HY : 0, hy_t = RP == 20 ? r_t - hzo_t * HY : r_t;
  const int xlds_t = (hzo_t * HY + hy_t) * PITCH + hx_t * ROWB + qd_t * 8;
  const int xx = x0 - 1 + hx_t;
  const bool vec = (Cin & 3) == 0;

  const uint4* wbase = a.wpk + (size_t)ctw * 64 + lane;
  const size_t wstep = (size_t)a.ct_total * 64;       // uint4 between consecutive (k-step, piece) slabs

  // a stage's tile: loads issued into registers one stage AHEAD (they fly during the previous stage's k-steps)
  u32x4 xr[NPASS];
  bool okv[NPASS];
  float4 nm = make_float4(0.f, 0.f, 0.f, 0.f), nr = make_float4(1.f, 1.f, 1.f, 1.f);
  auto issue_tile = [&](const int s) {
    const int cx_t = s * 4 * NQ + qd_t * 4;
    const bool xok = x_act && xx >= 0 && xx < W && cx_t < Cin;
    const unsigned xterm = ((unsigned)xx * (unsigned)Cin + (unsigned)cx_t) * 4u;
    const unsigned rowb = (unsigned)(W * Cin) * 4u;
    if (NORM && cx_t < Cin) {                          // (Cin % 4 == 0 whenever NORM: checked on the host)
      nm = *reinterpret_cast<const float4*>(a.in_mean + b * Cin + cx_t);
      nr = *reinterpret_cast<const float4*>(a.in_rstd + b * Cin + cx_t);
    }
    unsigned offv[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row0 = i * RP;
      const int z = z0 - 1 + row0 / HY + hzo_t, yy = y0 - 1 + row0 % HY + hy_t;
      okv[i] = xok && z >= 0 && z < D && yy >= 0 && yy < H;
      offv[i] = (unsigned)((b * D + z) * H + yy) * rowb + xterm;
    }
    if (vec) {                                         // (ONE uniform branch around the whole batch: a branch per item made
#pragma unroll                                         //  hipcc wait vmcnt(0) between the loads, conv3d_wtr.hip)
      for (int i = 0; i < NPASS; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, okv[i] ? offv[i] : Q_OOB, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        u32x4 v;
        v[0] = __builtin_amdgcn_raw_buffer_load_b32(rx, okv[i] ? offv[i] : Q_OOB, 0, 0);
        v[1] = __builtin_amdgcn_raw_buffer_load_b32(rx, okv[i] && cx_t + 1 < Cin ? offv[i] + 4 : Q_OOB, 0, 0);
        v[2] = __builtin_amdgcn_raw_buffer_load_b32(rx, okv[i] && cx_t + 2 < Cin ? offv[i] + 8 : Q_OOB, 0, 0);
        v[3] = __builtin_amdgcn_raw_buffer_load_b32(rx, okv[i] && cx_t + 3 < Cin ? offv[i] + 12 : Q_OOB, 0, 0);
        xr[i] = v;
      }
    }
  };
  issue_tile(0);
  float xsc = Q_F16_XSCALE, osc = 1.f / (Q_F16_XSCALE * Q_F16_WSCALE);      // NP 2: operand scale and the accumulator's way back
  if constexpr (NP == 2) {
    if (a.amax) {
      float m = a.amax[lane * MODET_AMAX_STRIDE];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      xsc = 1.f; osc = 1.f / Q_F16_WSCALE;
      if (m > 0.f && m < __builtin_huge_valf()) {
        int e;
        (void)frexpf(m, &e);
        int sh = 15 - e;
        sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
        xsc = ldexpf(1.f, sh); osc = ldexpf(1.f, -sh) * (1.f / Q_F16_WSCALE);
      }
    }
  }

  for (int s = 0; s < a.nstage; ++s) {
    // the first PF k-steps' weights of the stage in flight beside the tile (ring of PF register sets: a k-step is ~100-200
    // clocks of MFMA per wave, an L2 hit several hundred: one k-step of lookahead left every k-step waiting on its weights)
    const uint4* wst = wbase + (size_t)s * KS * NP * wstep;
    uint4 wq[PF][NP][CT];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int n = 0; n < CT; ++n)
          if (j < KS) wq[j][p][n] = wst[((size_t)j * NP + p) * wstep + n * 64];
    if (s > 0) __syncthreads();                        // every wave is done reading the previous stage's image
    if (x_act) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        float f0 = __uint_as_float(xr[i][0]), f1 = __uint_as_float(xr[i][1]), f2 = __uint_as_float(xr[i][2]), f3 = __uint_as_float(xr[i][3]);
        if (NORM) {                                    // zero padding applies to the NORMALISED tensor: out-of-volume stays 0
          f0 = okv[i] ? lrelu((f0 - nm.x) * nr.x) : 0.f; f1 = okv[i] ? lrelu((f1 - nm.y) * nr.y) : 0.f;
          f2 = okv[i] ? lrelu((f2 - nm.z) * nr.z) : 0.f; f3 = okv[i] ? lrelu((f3 - nm.w) * nr.w) : 0.f;
        }
        unsigned char* dst = lds + xlds_t + i * RP * PITCH;
        if constexpr (NP == 2) {
          unsigned h0, l0, h1, l1;
          split2h_pk(f0 * xsc, f1 * xsc, h0, l0);
          split2h_pk(f2 * xsc, f3 * xsc, h1, l1);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + XPL) = make_uint2(l0, l1);
        } else {
          unsigned h0, m0, l0, h1, m1, l1;
          split3_pk(f0, f1, h0, m0, l0);
          split3_pk(f2, f3, h1, m1, l1);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + XPL) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(dst + 2 * XPL) = make_uint2(l0, l1);
        }
      }
    }
    __syncthreads();
    if (s + 1 < a.nstage) issue_tile(s + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-steps: a lane's 8 k = chunks 8 ks + 2 kg, + 1 (table offsets), two 8-byte reads per piece and voxel tile
    // fully unrolled (KS is a constant <= 14): every condition below is a compile-time one, so the waits in front of the
    // MFMAs are exact vmcnt(n) counts -- with a run-time trip count hipcc fell back to vmcnt(0) at the head of each group
#pragma unroll
    for (int ks0 = 0; ks0 < KS; ks0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int ks = ks0 + j;
        if (ks < KS) {
          const int2 co2 = *reinterpret_cast<const int2*>(tab + 8 * ks + 2 * kg);
#pragma unroll
          for (int v = 0; v < VT; ++v) {
            bf16x8 xf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const uint2 lo = *reinterpret_cast<const uint2*>(lds + p * XPL + voff[v] + co2.x);
              const uint2 hi = *reinterpret_cast<const uint2*>(lds + p * XPL + voff[v] + co2.y);
              const u32x4 q4 = {lo.x, lo.y, hi.x, hi.y};
              xf[p] = __builtin_bit_cast(bf16x8, q4);
            }
#define MMQ(PW, PX)                                                                                                           \
            _Pragma("unroll") for (int n = 0; n < CT; ++n)                                                                    \
              acc[v][n] = q_mma<NP == 2>(__builtin_bit_cast(bf16x8, wq[j][PW][n]), xf[PX], acc[v][n]);
            if constexpr (NP == 3) { MMQ(2, 0) MMQ(0, 2) MMQ(1, 1) }
            MMQ(1, 0) MMQ(0, 1) MMQ(0, 0)
#undef MMQ
          }
          if (ks + PF < KS) {                          // this slot's next use
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
              for (int n = 0; n < CT; ++n) wq[j][p][n] = wst[((size_t)(ks + PF) * NP + p) * wstep + n * 64];
          }
          __builtin_amdgcn_sched_barrier(0);           // (the scheduler otherwise sinks these loads to their use, PF k-steps on)
        }
      }
    }
  }

  // ---- epilogue: D is [cout][voxel]: this lane holds voxel vj and the 4 CONSECUTIVE couts 4 kg .. + 3 of each of its cout tiles
#pragma unroll
  for (int n = 0; n < CT; ++n) {
    float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    const int co = (ctw + n) * 16 + kg * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, kv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (co + j < Cout) {
        if (a.bias) bv[j] = a.bias[co + j];
        if (STATS == 1) kv[j] = a.shift[b * Cout + co + j];
      }
    }
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      const int vt = vg * VT + v;
      const int z = z0 + (vt >> 2), yy = y0 + 2 * (vt & 3) + (vj >> 3), xo = x0 + (vj & 7);
      const bool ok = z < D && yy < H && xo < W;
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = NP == 2 ? fmaf(acc[v][n][j], osc, bv[j]) : acc[v][n][j] + bv[j];
        if (STATS == 1 && ok && co + j < Cout) { const float e = o[j] - kv[j]; sx[j] += e; sq[j] = fmaf(e, e, sq[j]); }
      }
      if constexpr (STATS == 2) {                      // (Cout % 4 == 0: checked on the host)
        if (ok && co < Cout) {
          const int64_t vox = (int64_t)((b * D + z) * H + yy) * W + xo;
          const float4 xr = *reinterpret_cast<const float4*>(a.xraw + vox * Cout + co);
          const float4 m4 = *reinterpret_cast<const float4*>(a.bmean + b * Cout + co);
          const float4 r4 = *reinterpret_cast<const float4*>(a.brstd + b * Cout + co);
          const float xv[4] = {xr.x, xr.y, xr.z, xr.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xh = (xv[j] - mv[j]) * rv[j];
            const float gg = o[j] * (xh > 0.f ? 1.f : LRELU_SLOPE);
            sx[j] += gg; sq[j] = fmaf(gg, xh, sq[j]);
          }
        }
      }
      if (ok) {
        float* dst = a.y + ((int64_t)((b * D + z) * H + yy) * W + xo) * Cout + co;
        if ((Cout & 3) == 0) {
          if (co < Cout) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < Cout) dst[j] = o[j];
        }
      }
    }
    if (STATS) {                                       // sum over the 16 voxel lanes; the waves sharing a cout tile meet in LDS
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { sx[j] += __shfl_xor(sx[j], o, 64); sq[j] += __shfl_xor(sq[j], o, 64); }
        if (vj == 0) {
          sred[((wave * CT + n) * 16 + kg * 4 + j) * 2] = sx[j];
          sred[((wave * CT + n) * 16 + kg * 4 + j) * 2 + 1] = sq[j];
        }
      }
    }
  }
  if (STATS) {
    __syncthreads();
    if (tid < CB * 16 * 2) {
      const int col = tid >> 1, which = tid & 1;       // col = (cout tile within the block) * 16 + cout within the tile
      const int cbt = col >> 4, c16 = col & 15;
      const int wc0 = cbt / CT, n = cbt - wc0 * CT;     // cout group of the wave, its n-th tile
      const int cog = (blockIdx.y * CB + cbt) * 16 + c16;
      if (cog < Cout) {
        float s4 = 0.f;
#pragma unroll
        for (int g = 0; g < 4 / WC; ++g) s4 += sred[(((g * WC + wc0) * CT + n) * 16 + c16) * 2 + which];     // waves g * WC + wc0, fixed order
        a.stats_rows[((int64_t)blockIdx.x * Cout + cog) * 2 + which] = s4;
      }
    }
  }
}

struct QPlan { int nq, nstage, wc, ct, cb, ct_total, ks, tiles_x, tiles_y, tiles_z; };
inline QPlan q_plan(int B, int D, int H, int W, int Cin, int Cout) {
  QPlan p;
  const int quads = (Cin + 3) / 4;
  p.nq = quads % 4 == 0 ? 4 : (quads % 3 == 0 ? 3 : (quads % 2 == 0 ? 2 : 1));
  p.nstage = quads / p.nq;
  p.ks = (27 * p.nq + 7) / 8;
  p.tiles_x = cdiv(W, TX); p.tiles_y = cdiv(H, TY); p.tiles_z = cdiv(D, TZ);
  const int ct = cdiv(Cout, 16);
  const int64_t tiles = (int64_t)B * p.tiles_x * p.tiles_y * p.tiles_z;
  // wave tiling (measured, tools/micro/convq_sweep.sh): two cout tiles per wave where that still leaves >= 2 workgroups per
  // CU (level 3), else one -- the wider tilings lose more to the shorter grid than the operand re-use gives back
  p.wc = 1;
  p.ct = (ct % 2 == 0 && tiles * (ct / 2) >= 512) ? 2 : 1;
  if (const char e = modet_tuning_env("MODET_CONVQ_TILING")) {      // "a".."f": (wc, ct) = (1,1) (1,2) (1,4) (2,1) (2,2) (4,1)
    const int t[6][2] = {{1, 1}, {1, 2}, {1, 4}, {2, 1}, {2, 2}, {4, 1}};
    if (e >= 'a' && e <= 'f') { p.wc = t[e - 'a'][0]; p.ct = t[e - 'a'][1]; }
  }
  p.cb = p.wc * p.ct;
  p.ct_total = cdiv(ct, p.cb) * p.cb;
  return p;
}
inline size_t q_wpk_elems(const QPlan& p) { return (size_t)p.nstage * p.ks * 3 * p.ct_total * 512; }

template <int NQ, int WC, int CT, int NP>
void q_launch_v(const QArgs& a, const dim3& grid, hipStream_t s) {
  if (a.xraw) {
    hipLaunchKernelGGL((conv_q_kernel<NQ, WC, CT, false, 2, NP>), grid, dim3(NTHR), 0, s, a);
  } else if (a.in_mean) {
    if (a.stats_rows) hipLaunchKernelGGL((conv_q_kernel<NQ, WC, CT, true, 1, NP>), grid, dim3(NTHR), 0, s, a);
    else hipLaunchKernelGGL((conv_q_kernel<NQ, WC, CT, true, 0, NP>), grid, dim3(NTHR), 0, s, a);
  } else {
    if (a.stats_rows) hipLaunchKernelGGL((conv_q_kernel<NQ, WC, CT, false, 1, NP>), grid, dim3(NTHR), 0, s, a);
    else hipLaunchKernelGGL((conv_q_kernel<NQ, WC, CT, false, 0, NP>), grid, dim3(NTHR), 0, s, a);
  }
}
template <int NQ, int NP>
void q_launch_n(const QArgs& a, const QPlan& p, const dim3& grid, hipStream_t s) {
  if (p.wc == 1 && p.ct == 1) q_launch_v<NQ, 1, 1, NP>(a, grid, s);
  else if (p.wc == 1 && p.ct == 2) q_launch_v<NQ, 1, 2, NP>(a, grid, s);
#ifdef MODET_TUNING
  else if (p.wc == 1 && p.ct == 4) q_launch_v<NQ, 1, 4, NP>(a, grid, s);
  else if (p.wc == 2 && p.ct == 1) q_launch_v<NQ, 2, 1, NP>(a, grid, s);
  else if (p.wc == 2 && p.ct == 2) q_launch_v<NQ, 2, 2, NP>(a, grid, s);
  else if (p.wc == 4 && p.ct == 1) q_launch_v<NQ, 4, 1, NP>(a, grid, s);
#endif
}

}  // namespace

// ---- internal interface for conv3d.hip (C++ linkage, not part of the ABI)
bool modetx_q_eligible(int B, int D, int H, int W, int Cin, int Cout) {
  const int64_t n = (int64_t)B * D * H * W;
  return Cin >= 2 && n * (Cin > Cout ? Cin : Cout) * 4 < 0x7fffffffLL && B <= 65535;
}
size_t modetx_q_ws_bytes(int Cin, int Cout) {
  // generous: any plan pads the chunk list to 8 per k-step (<= 14 k-steps per 16 channels) and Cout to 64
  return (size_t)cdiv(Cin, 4) * 4 * 27 * 2 * ((Cout + 63) / 64 * 64) * 3 * sizeof(unsigned short) + 65536;
}
// rows [B][tiles][Cout][2] of the data gradient's InstanceNorm-backward statistics (STATS 2); Cout = the gradient's channels
size_t modetx_q_bst_rows_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const QPlan p = q_plan(B, D, H, W, Cin, Cout);
  return (size_t)B * p.tiles_x * p.tiles_y * p.tiles_z * Cout * 2 * sizeof(float);
}
size_t modetx_q_stats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const QPlan p = q_plan(B, D, H, W, Cin, Cout);
  const size_t tiles = (size_t)p.tiles_x * p.tiles_y * p.tiles_z;
  return ((size_t)B * Cout + (size_t)B * tiles * Cout * 2) * sizeof(float);
}
// stats != null: [B][Cout] shift header (filled by the caller's shift kernel) followed by the rows [B][tiles][Cout][2]
int modetx_q_conv(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats,
                  const float* in_mean, const float* in_rstd, int B, int D, int H, int W, int Cin, int Cout, int mode,
                  hipStream_t s, const float* amax, const float* xraw, const float* bmean, const float* brstd, float* bst_rows, bool x_free) {
  const QPlan p = q_plan(B, D, H, W, Cin, Cout);
  unsigned short* wpk = (unsigned short*)ws;
  // forward: two f16 pieces when the caller vouches for the input's range (an activation; the lazily normalised input is one by
  // construction), else bf16x3 (fp32's range); data gradient: f16 when the caller knows max |d_y|, else bf16x3
  const bool f16 = Q_F16 && (int64_t)D * H * W < (1ll << 24) && (mode == 0 ? (!x_free || in_mean != nullptr) : amax != nullptr);
  const int np = f16 ? 2 : 3;
  // CoutP = 16 ct makes the arena's size formula (nstage * ksteps * CoutP * 32 * npiece elements) this packing's size
  const PackBKey key{w, Cin, Cout, p.ct_total * 16, p.nq, p.nstage, p.ks, mode, np, 4};
  const unsigned short* pre = nullptr;
  if (step) {
    std::lock_guard<std::mutex> lk(step->mu);
    if (step->active) {
      for (size_t i = 0; i < step->bjobs.size(); ++i)
        if (step->bjobs[i] == key) { pre = step->barena + step->boff[i]; break; }
    } else if (step->recording) {
      bool seen = false;
      for (const PackBKey& j : step->bjobs) seen = seen || j == key;
      if (!seen) step->bjobs.push_back(key);
    }
  }
  if (pre) wpk = const_cast<unsigned short*>(pre);
  else {
    const QPackJob J{w, wpk, Cin, Cout, p.nq, p.nstage, p.ks, p.ct_total, mode, np};
    const int total = p.nstage * p.ks * p.ct_total * 512;
    hipLaunchKernelGGL(q_pack_kernel, dim3(cdiv(total, 256) > 256 ? 256 : cdiv(total, 256)), dim3(256), 0, s, J);
  }
  QArgs a{x, (const uint4*)wpk, bias, y, in_mean, in_rstd, xraw ? bst_rows : (stats ? stats + (size_t)B * Cout : nullptr), stats,
          xraw, bmean, brstd, B, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, p.nstage, p.ct_total, (f16 && mode == 1) ? amax : nullptr};
  const dim3 grid(B * p.tiles_x * p.tiles_y * p.tiles_z, p.ct_total / p.cb);
#define Q_GO(NP_) do { \
    if (p.nq == 4) q_launch_n<4, NP_>(a, p, grid, s); \
    else if (p.nq == 3) q_launch_n<3, NP_>(a, p, grid, s); \
    else if (p.nq == 2) q_launch_n<2, NP_>(a, p, grid, s); \
    else q_launch_n<1, NP_>(a, p, grid, s); } while (0)
  if (f16) Q_GO(2); else Q_GO(3);
#undef Q_GO
  return modet_launch_status();
}
// the recorded packing jobs with layout 4 belong to this file (called from modetx_x3_prepack_begin's chain)
void modetx_q_prepack_begin(modet_step_ctx* c, hipStream_t stream) {
  std::vector<PackBKey> jobs;
  std::vector<size_t> off;
  unsigned short* arena;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    jobs = c->bjobs; off = c->boff; arena = c->barena;
  }
  QPackTable t;
  t.n = 0;
  auto go = [&]() {
    if (t.n) hipLaunchKernelGGL(q_pack_many_kernel, dim3(32, t.n), dim3(256), 0, stream, t);
    t.n = 0;
  };
  for (size_t i = 0; i < jobs.size(); ++i) {
    const PackBKey& k = jobs[i];
    if (k.layout != 4) continue;
    t.job[t.n++] = QPackJob{k.w, arena + off[i], k.Cin, k.Cout, k.CK, k.nstage, k.ksteps, k.CoutP / 16, k.mode, k.npiece};
    if (t.n == QPACK_MAX_JOBS) go();
  }
  go();
}
