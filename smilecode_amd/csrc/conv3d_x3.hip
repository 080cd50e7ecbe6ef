// 3x3x3 / stride 1 / zero-pad 1 convolution of the FEW-CHANNEL, FULL-RESOLUTION layers (pyramid levels 1-2: Cin, Cout <= 16)
// in fp32 accuracy on gfx950's bf16 matrix pipe ("bf16x3"), as a z-marching kernel.
//   reference call sites: nn.Conv3d in ConvBlock / ConvInsBlock, ModeT/models.py:127,:144 (arithmetic lives in ATen/MIOpen
//   there; these layers are 43 % of the forward FLOPs of the path, SURVEY.md section 7 "hard parts").
//
// Arithmetic.  Every fp32 operand is split into three bf16 pieces, x = hi + mid + lo exactly to 2^-24 (8 significand bits
// each, the remainders x - hi and (x - hi) - mid are exact in fp32); a product a*b is evaluated as the six piece products
// of total order <= 2, each EXACT in the fp32 accumulator of v_mfma_f32_16x16x32_bf16; the three dropped terms are
// <= 3 * 2^-24 |a b| -- the class of the single rounding of an fp32 FMA.  Six bf16 MFMAs (16 cycles each per SIMD) replace
// eight passes of the exact-f32 MFMA (32 cycles each): 2.7x less matrix-pipe time, and a third of the matrix-pipe power
// (the exact-f32 kernels run this chip into its package power limit: s_memtime shows ~1.5 GHz inside them on random data).
//
// Structure.  The exact-f32 kernel (conv3d.hip) stages a 4x8x16 tile with a one-voxel halo: 2.1x the tile's voxels pass
// through the split/LDS path and L2 (measured 1.9x the algorithmic HBM traffic).  Here a workgroup owns a (TY x 16) column
// of (y, x) and MARCHES ALONG z over a chunk of planes.  Each input plane is loaded from HBM once per column (halo 1.27x),
// split once, written to LDS once, and consumed once: its k-steps (the 3 x (P+2) taps of the plane x channels) feed the
// THREE output planes it contributes to (dz = 0, 1, 2), whose accumulators rotate through registers -- so one set of
// LDS operand reads serves 18 MFMAs, the weights (three pieces of 3 dz x NS k-steps) live in registers or LDS for the
// whole march, and LDS holds just two planes (double buffer), not a halo'd 3-D tile.  Software pipeline per plane:
// MFMAs on plane p | split + LDS write of plane p+1 (already in registers) | barrier | global loads of plane p+2.
// ROW PACKING (P = 2, Cout <= 8): the 16 MFMA rows are (row p in {0,1}) x (8 couts); the taps of a plane become (dyp in 0..3,
// dx): 12 taps x 8 channels = 3 k-steps with no padding (the exact-f32 kernel pays 25 % zero work for the same trick).
// dgrad is the same kernel on flipped / transposed weights.
#include "common.h"
#include "step_ctx.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 256;
constexpr int TX = 16, HX = TX + 2;

__device__ __forceinline__ unsigned short to_bf16(float a) {               // round to nearest even
  const __bf16 x = (__bf16)a;
  return __builtin_bit_cast(unsigned short, x);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = to_bf16(x);
  const float r1 = x - bf16_to_f32(hi);
  mid = to_bf16(r1);
  const float r2 = r1 - bf16_to_f32(mid);
  lo = to_bf16(r2);
}
// ---- TWO f16 pieces (round 5: the FORWARD launches).  x = hi + lo with hi = f16(x), lo = f16(x - hi): the residual is exact
// in fp32 and |lo| <= 2^-11 |x|, so the pair carries x to 2^-22 |x| (to 2^-25 absolute where lo is subnormal in f16, |x| <
// 0.125 -- gfx950's f16 MFMA keeps subnormal inputs: tools/micro/f16_denorm_probe.hip), and a product is the THREE piece
// products hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-22 |a b|): half the matrix-pipe work of the six bf16 products.  The
// price is range -- f16 overflows at 65 504 and the split degrades below 6e-5 -- so this form serves the FORWARD convolutions
// only, whose operands are images in [0,1], LeakyReLU(InstanceNorm(.)) activations and weights; gradients (any magnitude) stay on
// three bf16 pieces.  Measured error against fp64 on the parity tests' data: 3e-7 of max|y| (bf16x3: 1e-7; the tests' bound 2e-5).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_f16(float u, float v) {               // two roundings to nearest even, packed
  typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const h16x2 t = __builtin_convertvector((f32x2){u, v}, h16x2);
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void unpk_f16(unsigned p, float& u, float& v) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
  const h16x2 t = __builtin_bit_cast(h16x2, p);
  u = (float)t[0]; v = (float)t[1];
}
__device__ __forceinline__ void split2_h(float x, unsigned short& hi, unsigned short& lo) {
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}
#ifndef X3_VARIANT
#define X3_VARIANT 0
#endif
#ifndef X3_F16_FWD
#define X3_F16_FWD 1                     // forward launches on two f16 pieces (0: three bf16 pieces everywhere)
#endif
// The low piece of a value below 0.125 is SUBNORMAL in f16 (absolute resolution 2^-25): conv weights (|w| <= 1 / sqrt(27 Cin) ~
// 0.05-0.1) would all sit there and carry only ~20 bits.  Both operands are therefore scaled by exact powers of two before the
// split -- weights x 2^8 (normal low pieces down to |w| = 5e-4, overflow only beyond |w| = 255), activations x 2^4 (normal down
// to 8e-3, overflow beyond 4 094: LeakyReLU(InstanceNorm(.)) is bounded by sqrt(V)) -- and the accumulator is scaled back by
// 2^-12 in the epilogue's fma with the bias: no rounding anywhere in the scaling.  (The 4-channel instantiation -- the layer
// behind ConvBlock 1 -> 4, the one forward launch whose input is not normalised -- leaves its activations unscaled.)
constexpr float X3_F16_WSCALE = 256.f, X3_F16_XSCALE = 16.f;
__device__ __forceinline__ unsigned pk_bf16(float u, float v) {             // v_cvt_pk_bf16_f32: two roundings to nearest even
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const bf16x2 t = __builtin_convertvector((f32x2){u, v}, bf16x2);
  return __builtin_bit_cast(unsigned, t);
}
// two floats -> the packed bf16 pairs of their three pieces (v_cvt_pk_bf16_f32 + two masks per piece)
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

// Global traffic goes through buffer descriptors of ONE plane (wave-uniform base, 32-bit per-lane byte offset): an offset at
// or past num_records loads 0 / drops the store -- the zero padding of the halo, the ragged tile edges and "no plane here"
// (num_records = 0) all without a branch or a select, and with an instruction count the s_waitcnt pass can reason about.
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using BufRsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned X3_OOB = 0x80000000u;             // planes are < 2 GiB (checked on the host)
__device__ __forceinline__ BufRsrc plane_rsrc(const float* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// geometry shared by the kernel and the weight packing
template <int CIN, int P>
struct X3Geo {
  static constexpr int NTAP = 3 * (P + 2);                                 // taps of ONE input plane: (dyp, dx)
  static constexpr int NCH = CIN == 4 ? (NTAP + 1) / 2 : NTAP * (CIN / 8); // 8-wide k chunks per plane
  static constexpr int NS = (NCH + 3) / 4;                                 // k-steps (4 chunks = 32 k each)
  static constexpr int CoP = 16 / P;                                       // cout slots per packed row
};

template <bool F16>
__device__ __forceinline__ f32x4 x3_mma(bf16x8 a, bf16x8 b, f32x4 c) {       // (F16: the 16-byte fragments hold f16)
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ weight packing
// wpk[((dz * NS + s) * npiece + piece) * 64 + lane] = 8 bf16: the MFMA A fragment (row m = lane & 15, k chunk c = 4 s + (lane >> 4)).
//   chunk -> (tap, channels):  CIN 8: tap = c, channels 0..7;  CIN 16: tap = c / 2, channels 8 (c & 1) ..+7;
//                              CIN 4: element j < 4: tap 2c, channel j;  j >= 4: tap 2c + 1, channel j - 4.
//   tap = dyp * 3 + dx;  row m = p * CoP + co holds W((dz, dyp - p, dx); channel -> co) when 0 <= dyp - p <= 2, else 0.
//   mode 0 (forward): w[co][ch][tap27]   w: (Cout, Cin, 27);   mode 1 (dgrad): w[ch][co][26 - tap27]   w: (Co = ch, Ci = co, 27)
struct X3PackJob { const float* w; unsigned short* wpk; int Cin, Cout, cin_t, P, mode, npiece; };   // npiece 3 (fp32 emulation, bf16) | 2 (the same on f16 pieces) | 1 (bf16 storage)
constexpr int X3PACK_MAX_JOBS = 40;
struct X3PackTable { X3PackJob job[X3PACK_MAX_JOBS]; int n; };

__device__ __forceinline__ void x3_pack_body(const X3PackJob& J, int i0, int stride) {
  const int CIN = J.cin_t, P = J.P;
  const int NTAP = 3 * (P + 2), NCH = CIN == 4 ? (NTAP + 1) / 2 : NTAP * (CIN / 8), NS = (NCH + 3) / 4, CoP = 16 / P;
  const int per_piece = 64 * 8, total = 3 * NS * per_piece;               // per (dz, s): 3 pieces x 512 elements
  for (int i = i0; i < total; i += stride) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const int s = (i >> 9) % NS, dz = (i >> 9) / NS;
    const int m = lane & 15, c = s * 4 + (lane >> 4);
    int tap, ch;
    if (CIN == 8) { tap = c; ch = j; }
    else if (CIN == 16) { tap = c >> 1; ch = (c & 1) * 8 + j; }
    else { tap = 2 * c + (j >> 2); ch = j & 3; }
    const int p = m / CoP, co = m % CoP;
    const int dyp = tap / 3, dx = tap % 3, dy = dyp - p;
    float v = 0.f;
    if (tap < NTAP && dy >= 0 && dy <= 2 && co < J.Cout && ch < J.Cin) {
      const int t27 = (dz * 3 + dy) * 3 + dx;
      v = J.mode == 0 ? J.w[((int64_t)co * J.Cin + ch) * 27 + t27] : J.w[((int64_t)ch * J.Cout + co) * 27 + 26 - t27];
    }
    unsigned short h, md, l;
    unsigned short* o = J.wpk + ((size_t)((dz * NS + s) * J.npiece) * 64 + lane) * 8 + j;
    if (J.npiece == 2) {                                                  // two f16 pieces (forward launches), weights x 2^8
      split2_h(v * X3_F16_WSCALE, h, l);
      o[0] = h; o[per_piece] = l;
      continue;
    }
    split3(v, h, md, l);
    o[0] = h;                                                             // npiece 1: the weight rounded to bf16
    if (J.npiece == 3) { o[per_piece] = md; o[2 * per_piece] = l; }
  }
}
__global__ void x3_pack_kernel(const X3PackJob J) { x3_pack_body(J, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }
__global__ void x3_pack_many_kernel(const X3PackTable t) {
  x3_pack_body(t.job[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
inline int x3_ns(int cin_t, int P) {
  const int ntap = 3 * (P + 2), nch = cin_t == 4 ? (ntap + 1) / 2 : ntap * (cin_t / 8);
  return (nch + 3) / 4;
}
inline size_t x3_wpk_elems(int cin_t, int P, int npiece = 3) { return (size_t)3 * x3_ns(cin_t, P) * npiece * 64 * 8; }   // bf16 elements

// ------------------------------------------------------------------------------------------------ forward / dgrad
struct X3Args {
  const void* x; const uint4* wpk; const float* bias; void* y;           // x, y: fp32, or bf16 in the storage variants
  const float* in_mean; const float* in_rstd;          // NORM: x is a raw ConvInsBlock output, LeakyReLU((x - mean) * rstd) on load
  float* stats_rows; const float* shift;               // STATS 1: [b][item][Cout][2] sums of (y - K), (y - K)^2;  K = shift[b][Cout]
  const float* xraw; const float* bmean; const float* brstd;   // STATS 2 (dgrad whose output is the gradient w.r.t. LeakyReLU(
                                                       // InstanceNorm(xraw))): rows of sum g, sum g*xhat, g = y * lrelu'(xhat)
  int D, H, W, Cin, Cout, tiles_x, tiles_y, nchunk, ZC, nitems, act;
  long long* dbg;                                      // MODET_TUNING builds: per-(workgroup, wave) cycle sums per phase
  const float* amax;                                   // two f16 pieces, x = a GRADIENT: amax[0] >= max |x| sets its scale (else null)
};
// scale of a tensor whose magnitude is only known at run time (a gradient): amax = the maximum over the slots the producer
// left (include/modet_hip.h: modet_instnorm_lrelu_bwd_amax); the scale is the power of two that takes it into [2^14, 2^15)
// -- f16 overflows at 65 504 -- and its inverse.  amax = m 2^e with m in [0.5, 1): scale 2^(15 - e).  Zero / non-finite: 1.
__device__ __forceinline__ void x3_dyn_scale(const float* amax, float& sc, float& inv) {
  float m = amax[(threadIdx.x & 63) * MODET_AMAX_STRIDE];                  // MODET_AMAX_SLOTS = 64 slots, one per lane
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  sc = 1.f; inv = 1.f;
  if (m > 0.f && m < __builtin_huge_valf()) {
    int e;
    (void)frexpf(m, &e);
    int sh = 15 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    sc = ldexpf(1.f, sh); inv = ldexpf(1.f, -sh);
  }
}
#ifdef MODET_TUNING
__device__ long long* g_x3_dbg = nullptr;
#define X3_T(i) { const long long now_ = clock64(); dsum[i] += now_ - tprev; tprev = now_; }
#else
#define X3_T(i)
#endif

// CIN 4 / 8: two workgroups per CU (<= 256 registers; the partner's MFMAs cover this wave's staging).  CIN 16: the weights
// alone are 180 registers (3 dz x 5 k-steps x 3 pieces), so ONE workgroup per CU with the whole register file; a plane then
// carries 360 MFMAs per wave against ~150 staging instructions, which a single wave per SIMD absorbs.
// NPC = 1 (BASELINE.json configs[4], bf16 STORAGE / fp32 accumulate): one bf16 piece per operand -- the contract of conv3d_bf16.hip
// (exact arithmetic on the bf16-rounded operands, one rounding of the output) in this kernel's structure.  IN16: x is bf16 in
// HBM (a staging item = 8 channels = 16 bytes goes to LDS untouched), else fp32 rounded while staged; OUT16: y is stored as bf16.
// A plane then carries 18 MFMAs per wave instead of 108: the kernel is HBM-bound (8->8: 32 bytes per voxel).
template <int CIN, int P, int TY, bool WLDS, bool NORM, int STATS, int NPC = 3, bool IN16 = false, bool OUT16 = false>
#ifndef X3_OCC3
#define X3_OCC3 0                        // tuning builds: three workgroups per CU for the two-f16-piece forms with < 16 channels
#endif
__global__ __launch_bounds__(NTHR, NPC == 1 ? ((P == 2 && CIN < 16) ? 3 : 2) : (CIN == 16 ? 1 : ((((X3_VARIANT & 8) && CIN == 8) || (X3_OCC3 && NPC == 2)) ? 3 : 2))) void conv_x3_kernel(const X3Args a) {
  using G = X3Geo<CIN, P>;
  constexpr int NS = G::NS, NTAP = G::NTAP;
  constexpr int HY = TY + 2, UNITS = TY / P, R = UNITS / 4;
  // one piece's plane [hy][hx][CIN] bf16 + 16 bytes that absorb the LDS writes of staging slots past the plane's end
  constexpr int PLANE_D = HY * HX * CIN * 2, PLANE_B = PLANE_D + 16, SLOT_B = NPC * PLANE_B;
  constexpr int EPI = IN16 ? 8 : 4;                                        // channels per 16-byte staging item
  constexpr int Q = CIN / EPI, NITEM = HY * HX * Q, NIT = (NITEM + NTHR - 1) / NTHR;
  // STATS 2 (8 / 16 input channels): the LOW piece of the weights lives in LDS, the other two in registers -- its A
  // fragments are read once per (plane, k-step) and feed the last MFMAs of the unit; frees 36 registers for the statistics
  // (X3_VARIANT bits 4 / 8, tuning builds: the low piece / the two low pieces in LDS for every fp32 instantiation with 8+ channels)
  constexpr int NPX = NPC == 1 ? 1 : NPC;                                  // pieces per operand in the product loops
  // f16 pieces: activation scale (4 input channels = the layer behind ConvBlock 1 -> 4, whose input is NOT normalised: unscaled,
  // so that it only overflows beyond 65 504) and the scale that takes the accumulator back
  constexpr float XSC = CIN == 4 ? 1.f : X3_F16_XSCALE, OSC = 1.f / (X3_F16_WSCALE * XSC);
  float xsc = XSC, osc = OSC;                                              // (a.amax: the input is a gradient, scaled by its maximum)
  if constexpr (NPC == 2) {
    if (a.amax) { float sc, inv; x3_dyn_scale(a.amax, sc, inv); xsc = sc; osc = inv * (1.f / X3_F16_WSCALE); }
  }
  constexpr int NPR = (NPC != 3 || WLDS || CIN < 8) ? NPC                  // pieces kept in registers
                      : ((X3_VARIANT & 8) && CIN == 8) ? 1 : ((STATS == 2 || (X3_VARIANT & 4)) ? 2 : 3);
  constexpr bool WLO = NPR < NPC;
  constexpr int WL_B = WLDS ? 3 * NS * NPC * 1024 : (WLO ? 3 * NS * (NPC - NPR) * 1024 : 16);
  static_assert(UNITS % 4 == 0, "row groups split over 4 waves");
  static_assert(Q >= 1 && NTHR % Q == 0, "a thread's staging items share one channel group");
  static_assert(NPC != 1 || !NORM, "the lazily normalised input exists for the fp32 forms only");
  static_assert(NPC == 1 || NPC == 2 || NPC == 3, "pieces per operand");
  static_assert(NPC == 1 || (!IN16 && !OUT16), "bf16 tensors belong to the one-piece form");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * SLOT_B];
  __shared__ __attribute__((aligned(16))) unsigned char wl[WL_B];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  // XCD-aware item order: workgroups land on XCD blockIdx % 8; give each XCD a contiguous range of items so that
  // columns sharing halo rows / planes share an L2
  const int per_xcd = (a.nitems + 7) >> 3;
  int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (item >= a.nitems) return;
  const int item_id = item;
  const int tx = item % a.tiles_x; item /= a.tiles_x;
  const int ty = item % a.tiles_y; item /= a.tiles_y;
  const int zc = item % a.nchunk;
  const int b = item / a.nchunk;
  const int x0 = tx * TX, y0 = ty * TY, zs = zc * a.ZC;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int nz = (zs + a.ZC <= D ? a.ZC : D - zs);                         // output planes of this item

  // ---- weights: A fragments of every (dz, k-step, piece), in registers (or LDS) for the whole march
  uint4 wreg[WLDS ? 1 : 3][WLDS ? 1 : NS][WLDS ? 1 : NPR];
  if constexpr (WLDS) {
    for (int i = tid; i < 3 * NS * NPC * 64; i += NTHR) reinterpret_cast<uint4*>(wl)[i] = a.wpk[i];
  } else {
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int pc = 0; pc < NPR; ++pc) wreg[dz][s][pc] = a.wpk[((dz * NS + s) * NPC + pc) * 64 + lane];
    if constexpr (WLO) {                                                   // wl[((dz*NS + s) * (NPC - NPR) + pc - NPR) * 64 + lane]
      for (int i = tid; i < 3 * NS * (NPC - NPR) * 64; i += NTHR) {
        const int ln = i & 63, j = i >> 6, pcl = j % (NPC - NPR), ds = j / (NPC - NPR);
        reinterpret_cast<uint4*>(wl)[i] = a.wpk[(ds * NPC + NPR + pcl) * 64 + ln];
      }
    }
  }

  // ---- staging map: item i = tid + j * NTHR of the halo'd plane [hy][hx][Q channel groups]
  unsigned goff[NIT];                                                      // byte offset inside the input plane, or out of bounds
  int loff[NIT];
  unsigned okmask = 0;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int i = tid + j * NTHR;
    const bool on = i < NITEM;
    const int v = on ? i / Q : 0, c4 = on ? i - v * Q : 0;
    const int hy = v / HX, hx = v - hy * HX;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool ok = on && yy >= 0 && yy < H && xx >= 0 && xx < W && c4 * EPI < Cin;
    goff[j] = ok ? (unsigned)(((yy * W + xx) * Cin + c4 * EPI) * (IN16 ? 2 : 4)) : X3_OOB;
    loff[j] = on ? (v * CIN + c4 * EPI) * 2 : PLANE_D;                     // slots past the plane's end write the pad
    okmask |= (ok ? 1u : 0u) << j;
  }
  float4 nm = make_float4(0.f, 0.f, 0.f, 0.f), nr = make_float4(1.f, 1.f, 1.f, 1.f);
  if constexpr (NORM) {
    const int cg = (tid % Q) * 4;
    if (cg < Cin) {
      nm = *reinterpret_cast<const float4*>(a.in_mean + b * Cin + cg);
      nr = *reinterpret_cast<const float4*>(a.in_rstd + b * Cin + cg);
    }
  }
  constexpr int ISZ = IN16 ? 2 : 4, OSZ = OUT16 ? 2 : 4;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + (int64_t)b * D * H * W * Cin * ISZ;
  const unsigned in_plane_bytes = (unsigned)H * W * Cin * ISZ, out_plane_bytes = (unsigned)H * W * Cout * OSZ;
  // XSETS register sets, plane q lives in set q % XSETS.  One set = the plane is loaded one iteration before it is split
  // (measured: the split then waits ~1 % of a wave's cycles for its loads; a second iteration of distance bought nothing
  // and its 24 registers are worth more to the MFMA loop's operand prefetch)
  constexpr int XSETS = 1;
  float4 xr[XSETS][NIT];
  bool xr_live[XSETS] = {};                                                // the set holds an in-volume plane
  auto load_plane = [&](auto set_c, int z) {
    constexpr int S = decltype(set_c)::value;
    xr_live[S] = z >= 0 && z < D;
    const BufRsrc rs = plane_rsrc(reinterpret_cast<const float*>(xb + (int64_t)(xr_live[S] ? z : 0) * in_plane_bytes),
                                  xr_live[S] ? in_plane_bytes : 0u);
#pragma unroll
    for (int j = 0; j < NIT; ++j) xr[S][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[j], 0, 0));
  };
  auto store_plane = [&](auto set_c, int slot) {
    constexpr int S = decltype(set_c)::value;
    unsigned char* sl = lds + slot * SLOT_B;
    if constexpr (NPC == 1) {
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const float4 t = xr[S][j];
        if constexpr (IN16) *reinterpret_cast<uint4*>(sl + loff[j]) = __builtin_bit_cast(uint4, t);     // 8 bf16, untouched
        else *reinterpret_cast<uint2*>(sl + loff[j]) = make_uint2(pk_bf16(t.x, t.y), pk_bf16(t.z, t.w));
      }
      return;
    }
    if constexpr (NPC == 2) {                                              // two f16 pieces: hi, then the exact residual
      unsigned hi2[NIT][2], lo2[NIT][2];
      float4 t[NIT];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        t[j] = xr[S][j];
        if constexpr (NORM) {
          const bool on = xr_live[S] && ((okmask >> j) & 1u);
          t[j].x = on ? lrelu((t[j].x - nm.x) * nr.x) : 0.f; t[j].y = on ? lrelu((t[j].y - nm.y) * nr.y) : 0.f;
          t[j].z = on ? lrelu((t[j].z - nm.z) * nr.z) : 0.f; t[j].w = on ? lrelu((t[j].w - nm.w) * nr.w) : 0.f;
        }
        t[j].x *= xsc; t[j].y *= xsc; t[j].z *= xsc; t[j].w *= xsc;
      }
#pragma unroll
      for (int j = 0; j < NIT; ++j) { hi2[j][0] = pk_f16(t[j].x, t[j].y); hi2[j][1] = pk_f16(t[j].z, t[j].w); }
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        float a0, a1, a2, a3;
        unpk_f16(hi2[j][0], a0, a1); unpk_f16(hi2[j][1], a2, a3);
        lo2[j][0] = pk_f16(t[j].x - a0, t[j].y - a1); lo2[j][1] = pk_f16(t[j].z - a2, t[j].w - a3);
      }
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        *reinterpret_cast<uint2*>(sl + loff[j]) = make_uint2(hi2[j][0], hi2[j][1]);
        *reinterpret_cast<uint2*>(sl + PLANE_B + loff[j]) = make_uint2(lo2[j][0], lo2[j][1]);
      }
      return;
    }
    // all NIT items stage by stage (2 NIT independent dependency chains side by side: the split is a chain of seven
    // dependent VALU operations per value pair, one item after the other leaves the VALU waiting on itself)
    float2 v[NIT][2];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      float4 t = xr[S][j];
      if constexpr (NORM) {                                                // zero padding stays zero (selects, no branch)
        const bool on = xr_live[S] && ((okmask >> j) & 1u);
        t.x = on ? lrelu((t.x - nm.x) * nr.x) : 0.f; t.y = on ? lrelu((t.y - nm.y) * nr.y) : 0.f;
        t.z = on ? lrelu((t.z - nm.z) * nr.z) : 0.f; t.w = on ? lrelu((t.w - nm.w) * nr.w) : 0.f;
      }
      v[j][0] = make_float2(t.x, t.y); v[j][1] = make_float2(t.z, t.w);
    }
    unsigned hi[NIT][2], mid[NIT][2], lo[NIT][2];
#if X3_VARIANT & 2
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) hi[j][h] = pk_bf16(v[j][h].x, v[j][h].y);
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[j][h].x -= __uint_as_float(hi[j][h] << 16); v[j][h].y -= __uint_as_float(hi[j][h] & 0xffff0000u);
      }
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) mid[j][h] = pk_bf16(v[j][h].x, v[j][h].y);
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[j][h].x -= __uint_as_float(mid[j][h] << 16); v[j][h].y -= __uint_as_float(mid[j][h] & 0xffff0000u);
      }
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) lo[j][h] = pk_bf16(v[j][h].x, v[j][h].y);
#else
#pragma unroll
    for (int j = 0; j < NIT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) split3_pk(v[j][h].x, v[j][h].y, hi[j][h], mid[j][h], lo[j][h]);
#endif
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      *reinterpret_cast<uint2*>(sl + loff[j]) = make_uint2(hi[j][0], hi[j][1]);
      *reinterpret_cast<uint2*>(sl + PLANE_B + loff[j]) = make_uint2(mid[j][0], mid[j][1]);
      *reinterpret_cast<uint2*>(sl + 2 * PLANE_B + loff[j]) = make_uint2(lo[j][0], lo[j][1]);
    }
  };

  // ---- operand addresses: B fragment = 8 k values of voxel (row group r, x = li): chunk c = 4 s + lk
  int coffA[NS], coffB[CIN == 4 ? NS : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = s * 4 + lk;
    if (CIN == 4) {
      int t0 = 2 * c, t1 = 2 * c + 1;
      if (t0 >= NTAP) t0 = 0;                                              // padded k: finite data times zero weights
      if (t1 >= NTAP) t1 = 0;
      coffA[s] = ((t0 / 3) * HX + t0 % 3) * CIN * 2;
      coffB[s] = ((t1 / 3) * HX + t1 % 3) * CIN * 2;
    } else {
      int tap = CIN == 16 ? c >> 1 : c;
      const int ch0 = CIN == 16 ? (c & 1) * 8 : 0;
      if (tap >= NTAP) tap = 0;
      coffA[s] = (((tap / 3) * HX + tap % 3) * CIN + ch0) * 2;
    }
  }
  int xoff[R];
#pragma unroll
  for (int r = 0; r < R; ++r) xoff[r] = ((P * (wave * R + r)) * HX + li) * CIN * 2;

  f32x4 acc[3][R];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[q][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- epilogue constants: this lane holds 4 consecutive couts co0.. of voxel (row, x0 + li) of each row group
  const int co0 = P == 2 ? (lk & 1) * 4 : lk * 4;
  const bool co_ok = co0 < Cout;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (co_ok && a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + co0);
  if (STATS == 1 && co_ok) k4 = *reinterpret_cast<const float4*>(a.shift + b * Cout + co0);
  // STATS 2: InstanceNorm-backward statistics of this launch's OUTPUT (a data gradient) against the raw tensor the norm
  // was applied to (same shape as the output): its float4 of the NEXT plane to be flushed is loaded one iteration ahead
  // (mean, rstd of the sample's 16 channel slots sit in LDS and are re-read by every flush: the 8->8 instantiation has no
  // registers to spare -- with them and a second operand buffer in registers it spilled 36 and ran at half speed)
  __shared__ __attribute__((aligned(16))) float bst_mr[STATS == 2 ? 32 : 1];
  if constexpr (STATS == 2) {
    if (tid < 32) {
      const int c = tid & 15;
      bst_mr[tid] = c < Cout ? (tid < 16 ? a.bmean[b * Cout + c] : a.brstd[b * Cout + c]) : 0.f;
    }
  }
  float4 xraw_nx[STATS == 2 ? R : 1];
#pragma unroll
  for (int r = 0; r < (STATS == 2 ? R : 1); ++r) xraw_nx[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* xrawb = STATS == 2 ? a.xraw + (int64_t)b * D * H * W * Cout : nullptr;
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + (int64_t)b * D * H * W * Cout * OSZ;
  unsigned soff[R];                                                        // byte offset of this lane's 4 couts in an output plane
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int u = wave * R + r;
    const int row = y0 + (P == 2 ? 2 * u + (lk >> 1) : u);
    soff[r] = (co_ok && x0 + li < W && row < H) ? (unsigned)(((row * W + x0 + li) * Cout + co0) * OSZ) : X3_OOB;
  }

  auto frag = [&](const unsigned char* sl, int pc, int r, int s) -> bf16x8 {
    const unsigned char* base = sl + pc * PLANE_B + xoff[r];
    if constexpr (CIN == 4) {
      const uint2 lo = *reinterpret_cast<const uint2*>(base + coffA[s]);
      const uint2 hi = *reinterpret_cast<const uint2*>(base + coffB[s]);
      return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
    } else {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + coffA[s]));
    }
  };
  auto wfrag = [&](int dz, int s, int pc) -> bf16x8 {
    if constexpr (WLDS) return __builtin_bit_cast(bf16x8, reinterpret_cast<const uint4*>(wl)[((dz * NS + s) * NPC + pc) * 64 + lane]);
    else if constexpr (WLO) {
      if (pc >= NPR) return __builtin_bit_cast(bf16x8, reinterpret_cast<const uint4*>(wl)[((dz * NS + s) * (NPC - NPR) + pc - NPR) * 64 + lane]);
      return __builtin_bit_cast(bf16x8, wreg[dz][s][pc < NPR ? pc : 0]);
    } else return __builtin_bit_cast(bf16x8, wreg[dz][s][pc]);
  };
#define X3_MM(ACC, WP, XP) ACC = x3_mma<NPC == 2>(w[WP], xf[XP], ACC)
  // one input plane (LDS slot) into the three output planes it touches; C = q % 3: dz -> accumulator (C - dz) mod 3
  auto compute = [&](auto cc, int slot, bool a0, bool a1, bool a2) {
    constexpr int C = decltype(cc)::value;
    constexpr int S0 = C, S1 = (C + 2) % 3, S2 = (C + 1) % 3;
    const unsigned char* sl = lds + slot * SLOT_B;
    if constexpr (NPC == 1) {
      // one product per (dz, k-step, row group); the B fragment of unit u+1 is read before the MFMAs of unit u
      bf16x8 xq[2];
      xq[0] = frag(sl, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NS * R; ++u) {
        const int s = u / R, r = u % R;
        if (u + 1 < NS * R) xq[(u + 1) & 1] = frag(sl, 0, (u + 1) % R, (u + 1) / R);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 xf = xq[u & 1];
        if (a0) acc[S0][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(0, s, 0), xf, acc[S0][r], 0, 0, 0);
        if (a1) acc[S1][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(1, s, 0), xf, acc[S1][r], 0, 0, 0);
        if (a2) acc[S2][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(2, s, 0), xf, acc[S2][r], 0, 0, 0);
      }
    } else if (a0 && a1 && a2) {
      // (k-step, row group) units in order; the three B fragments of unit u+1 are read from LDS before the 18 MFMAs of unit u
      constexpr int NXQ = STATS == 2 ? 1 : 2;                  // (STATS 2: no register room for the second buffer)
      bf16x8 xq[NXQ][NPX];
#pragma unroll
      for (int pc = 0; pc < NPX; ++pc) xq[0][pc] = frag(sl, pc, 0, 0);
#pragma unroll
      for (int u = 0; u < NS * R; ++u) {
        const int s = u / R, r = u % R;
        if (NXQ == 1 && u > 0) {
#pragma unroll
          for (int pc = 0; pc < NPX; ++pc) xq[0][pc] = frag(sl, pc, r, s);
        }
        if (NXQ == 2 && u + 1 < NS * R) {
#pragma unroll
          for (int pc = 0; pc < NPX; ++pc) xq[(u + 1) % NXQ][pc] = frag(sl, pc, (u + 1) % R, (u + 1) / R);
        }
        // keep the reads up here: left alone the scheduler sinks them to their first use to save registers and every
        // unit then starts with an exposed LDS round trip (seen in the ISA: ds_read, s_waitcnt lgkmcnt(0), v_mfma)
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 w0[NPX], w1[NPX], w2[NPX];
#pragma unroll
        for (int pc = 0; pc < NPX; ++pc) { w0[pc] = wfrag(0, s, pc); w1[pc] = wfrag(1, s, pc); w2[pc] = wfrag(2, s, pc); }
        const bf16x8* xf = xq[u % NXQ];
        // small terms first; the three accumulators alternate so that back-to-back MFMAs are independent
#define X3_ALL(WP, XP) { bf16x8* w = w0; X3_MM(acc[S0][r], WP, XP); } { bf16x8* w = w1; X3_MM(acc[S1][r], WP, XP); } { bf16x8* w = w2; X3_MM(acc[S2][r], WP, XP); }
        if constexpr (NPC == 2) { X3_ALL(1, 0) X3_ALL(0, 1) X3_ALL(0, 0) }                                  // f16 pair: three products
        else if constexpr (WLO) { X3_ALL(0, 2) X3_ALL(1, 1) X3_ALL(1, 0) X3_ALL(0, 1) X3_ALL(0, 0) X3_ALL(2, 0) }   // LDS-resident piece last
        else { X3_ALL(2, 0) X3_ALL(0, 2) X3_ALL(1, 1) X3_ALL(1, 0) X3_ALL(0, 1) X3_ALL(0, 0) }
#undef X3_ALL
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          bf16x8 xf[NPX];
#pragma unroll
          for (int pc = 0; pc < NPX; ++pc) xf[pc] = frag(sl, pc, r, s);
#define X3_ONE(DZ, SL) { bf16x8 w[NPX]; for (int pc = 0; pc < NPX; ++pc) w[pc] = wfrag(DZ, s, pc); \
            if constexpr (NPC == 2) { X3_MM(acc[SL][r], 1, 0); X3_MM(acc[SL][r], 0, 1); X3_MM(acc[SL][r], 0, 0); } \
            else { X3_MM(acc[SL][r], NPX - 1, 0); X3_MM(acc[SL][r], 0, NPX - 1); X3_MM(acc[SL][r], 1, 1); X3_MM(acc[SL][r], 1, 0); X3_MM(acc[SL][r], 0, 1); X3_MM(acc[SL][r], 0, 0); } }
          if (a0) X3_ONE(0, S0)
          if (a1) X3_ONE(1, S1)
          if (a2) X3_ONE(2, S2)
#undef X3_ONE
        }
      }
    }
  };
#undef X3_MM
  // output plane z is complete, its accumulator is SL: + bias, statistics, activation, one 16-byte store per row group.
  // live = false (no finished plane yet): the same instructions run against an empty descriptor and store nothing.
  auto flush = [&](auto sl_c, int z, bool live, bool next_live) {
    constexpr int SL = decltype(sl_c)::value;
    const BufRsrc rs = plane_rsrc(reinterpret_cast<const float*>(yb + (int64_t)(live ? z : 0) * out_plane_bytes), live ? out_plane_bytes : 0u);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f32x4 v = acc[SL][r];
      acc[SL][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float o[4];
      if constexpr (NPC == 2) {                              // (the f16 operands were scaled by 2^4 x 2^8)
        o[0] = fmaf(v[0], osc, b4.x); o[1] = fmaf(v[1], osc, b4.y); o[2] = fmaf(v[2], osc, b4.z); o[3] = fmaf(v[3], osc, b4.w);
      } else {
        o[0] = v[0] + b4.x; o[1] = v[1] + b4.y; o[2] = v[2] + b4.z; o[3] = v[3] + b4.w;
      }
      if (STATS == 1) {
        const bool on = live && soff[r] != X3_OOB;
        const float e0 = on ? o[0] - k4.x : 0.f, e1 = on ? o[1] - k4.y : 0.f, e2 = on ? o[2] - k4.z : 0.f, e3 = on ? o[3] - k4.w : 0.f;
        sx[0] += e0; sx[1] += e1; sx[2] += e2; sx[3] += e3;
        sq[0] = fmaf(e0, e0, sq[0]); sq[1] = fmaf(e1, e1, sq[1]); sq[2] = fmaf(e2, e2, sq[2]); sq[3] = fmaf(e3, e3, sq[3]);
      }
      if constexpr (STATS == 2) {
        const bool on = live && soff[r] != X3_OOB;
        const float4 xr = xraw_nx[r];                          // loaded by the previous iteration's flush for THIS plane
        const float4 bm4 = *reinterpret_cast<const float4*>(bst_mr + (co_ok ? co0 : 0));
        const float4 br4 = *reinterpret_cast<const float4*>(bst_mr + 16 + (co_ok ? co0 : 0));
        const float xv[4] = {xr.x, xr.y, xr.z, xr.w}, mv[4] = {bm4.x, bm4.y, bm4.z, bm4.w}, rv[4] = {br4.x, br4.y, br4.z, br4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - mv[j]) * rv[j];
          const float gg = on ? o[j] * (xh > 0.f ? 1.f : LRELU_SLOPE) : 0.f;
          sx[j] += gg; sq[j] = fmaf(gg, xh, sq[j]);
        }
      }
      if (a.act) { o[0] = lrelu(o[0]); o[1] = lrelu(o[1]); o[2] = lrelu(o[2]); o[3] = lrelu(o[3]); }
      if constexpr (OUT16) {                                               // statistics above: of the fp32 values, as conv3d_bf16.hip
        const u32x2 o2 = {pk_bf16(o[0], o[1]), pk_bf16(o[2], o[3])};
        __builtin_amdgcn_raw_buffer_store_b64(o2, rs, (int)soff[r], 0, 0);
      } else {
        const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), rs, (int)soff[r], 0, 0);
      }
    }
    if constexpr (STATS == 2) {                                          // the raw tensor's values for the next plane to be flushed
      const BufRsrc rx = plane_rsrc(xrawb + (int64_t)(next_live ? z + 1 : 0) * H * W * Cout, next_live ? out_plane_bytes : 0u);
#pragma unroll
      for (int r = 0; r < R; ++r) xraw_nx[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)soff[r], 0, 0));
    }
  };

  // ---- the march: q-th input plane = zs - 1 + q, q = 0 .. nz + 1.  One iteration:
  //   split + LDS write of plane q+1 (registers loaded two iterations ago) | stores of output plane q-3 (final since the
  //   previous iteration) | global loads of plane q+3 | MFMAs of plane q | barrier.
  // Every memory op younger than the loads the split waits for is counted by the compiler (straight-line buffer ops, no
  // divergent branches), so the wait is an exact vmcnt(n), never a drain of this iteration's stores.
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  const int nq = nz + 2;
  load_plane(I0{}, zs - 1);
  if (WLDS || WLO) __syncthreads();
  store_plane(I0{}, 0);
  load_plane(I0{}, zs);
  __syncthreads();
#ifdef MODET_TUNING
  long long dsum[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#endif
  // The loop runs whole groups of three iterations (the accumulator / register-set rotation is compile-time) with
  // IDENTICAL memory-op structure -- [wait][R stores][NIT loads] -- in every iteration, also the padded ones past the last
  // plane (empty descriptors, no MFMAs): the s_waitcnt pass then derives an exact vmcnt(n) for the split's wait.
  auto body = [&](auto cc, int q) {
    constexpr int C = decltype(cc)::value;
#if X3_VARIANT & 1
    __builtin_amdgcn_s_setprio(2);                                         // staging / epilogue VALU ahead of the partner wave's MFMAs
#endif
#ifdef MODET_TUNING
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                        // the split's loads (and two of the next set): wait time alone
    X3_T(5)
#endif
    if (q + 1 < nq) store_plane(I0{}, (q + 1) & 1);
    X3_T(2)
    flush(cc, zs + q - 3, q >= 3 && q <= nq, q + 1 >= 3 && q + 1 <= nq);   // slot C: output q-3, about to be re-used for output q
    X3_T(1)
    // plane q+2 into the registers the split just emptied; past the last plane: an empty descriptor (zeros, no traffic)
    load_plane(I0{}, q + 2 < nq ? zs + q + 1 : -1);
    X3_T(4)
#if X3_VARIANT & 1
    __builtin_amdgcn_s_setprio(0);
#endif
    compute(cc, q & 1, q <= nz - 1, q >= 1 && q <= nz, q >= 2 && q <= nz + 1);
    X3_T(0)
    __syncthreads();
    X3_T(3)
  };
  for (int q = 0; q <= nq; q += 3) {                                       // iteration nq only flushes the last output plane
    body(I0{}, q);
    body(I1{}, q + 1);
    body(I2{}, q + 2);
  }

#ifdef MODET_TUNING
  if (g_x3_dbg && lane == 0) {
    long long* o = g_x3_dbg + ((int64_t)blockIdx.x * 4 + wave) * 6;
    for (int i = 0; i < 6; ++i) o[i] = dsum[i];
  }
#endif
  if constexpr (STATS != 0) {
    // lanes sharing a channel group: all li, and for P == 2 both rows (lk >> 1); then the 4 waves through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { sx[j] += __shfl_xor(sx[j], o, 64); sq[j] += __shfl_xor(sq[j], o, 64); }
      if (P == 2) { sx[j] += __shfl_xor(sx[j], 32, 64); sq[j] += __shfl_xor(sq[j], 32, 64); }
    }
    __syncthreads();                                                       // every wave is done with the plane buffers
    float* sred = reinterpret_cast<float*>(lds);                           // [wave][16 channels][2]
    if (li == 0 && (P == 1 || lk < 2)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sred[(wave * 16 + co0 + j) * 2] = sx[j]; sred[(wave * 16 + co0 + j) * 2 + 1] = sq[j]; }
    }
    __syncthreads();
    if (tid < 2 * Cout) {
      float t = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) t += sred[w4 * 32 + tid];
      const int rows_per_b = a.tiles_x * a.tiles_y * a.nchunk;
      a.stats_rows[((int64_t)b * rows_per_b + (item_id - b * rows_per_b)) * Cout * 2 + tid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
struct X3Plan { int cin_t, P, ty, wlds, tiles_x, tiles_y, nchunk, zc, nitems; };

inline bool x3_shape_ok(int Cin, int Cout) { return Cin % 4 == 0 && Cin >= 4 && Cin <= 16 && Cout % 4 == 0 && Cout >= 4 && Cout <= 16; }

inline X3Plan x3_plan(int B, int D, int H, int W, int Cin, int Cout, int npc = 3) {
  X3Plan p;
  p.cin_t = Cin <= 4 ? 4 : (Cin <= 8 ? 8 : 16);
  // CIN 16 runs unpacked (P = 1) also for Cout <= 8: its packed form needs 6 k-steps = 216 weight registers (one piece: 18)
  p.P = (Cout <= 8 && (p.cin_t < 16 || npc == 1)) ? 2 : 1;
  // 16 rows = 8 packed row pairs, or 8 plain rows: two row groups per wave; CIN 16 / one piece: 16 plain rows, four per wave
  p.ty = (p.P == 2 || p.cin_t == 16 || npc == 1) ? 16 : 8;
  p.wlds = 0;
  const int slots = npc == 1 ? ((p.P == 2 && p.cin_t < 16) ? 768 : 512)                                      // resident workgroups
                             : (p.cin_t == 16 ? 256 : ((((X3_VARIANT & 8) && p.cin_t == 8) || X3_OCC3) ? 768 : 512));
  p.tiles_x = cdiv(W, TX);
  p.tiles_y = cdiv(H, p.ty);
  // z chunks: enough workgroups to fill 256 CUs x 2 several times over (the dispatcher balances them), but chunks long
  // enough that the two extra halo planes and the pipeline prologue stay small
  const int cols = B * p.tiles_x * p.tiles_y;
  int best = 1;
  double best_cost = 1e30;
  for (int n = 1; n <= D; ++n) {
    const int zc = cdiv(D, n);
    if (zc < 4 && n > 1) break;
    const int64_t items = (int64_t)cols * cdiv(D, zc);
    const int64_t rounds = (items + slots - 1) / slots;
    const double cost = (double)rounds * (zc + 2.5);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
  }
  p.zc = cdiv(D, best);
  p.nchunk = cdiv(D, p.zc);
  p.nitems = cols * p.nchunk;
  return p;
}

// packed weights of this launch: the step context's arena (prepacked) or `ws` (packed here)
inline const uint4* x3_weights(modet_step_ctx* step, const float* w, void* ws, int Cin, int Cout, int mode, int npc, const X3Plan& p,
                               hipStream_t s) {
  unsigned short* wpk = (unsigned short*)ws;
  const PackBKey key{w, Cin, Cout, 16, p.cin_t, 3, x3_ns(p.cin_t, p.P), mode, npc, 1 + p.P};   // nstage = the 3 dz
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
  else hipLaunchKernelGGL(x3_pack_kernel, dim3(8), dim3(256), 0, s, X3PackJob{w, wpk, Cin, Cout, p.cin_t, p.P, mode, npc});
  return (const uint4*)wpk;
}

template <bool NORM, bool STATS>
int x3_launch(modet_step_ctx* step, const X3Args& a0, const float* w, void* ws, int B, int mode, const X3Plan& p, hipStream_t s, bool x_free) {
  X3Args a = a0;
  // forward: two f16 pieces, three products (see split2_h) when the caller vouches for the input's range (x_free == false: an
  // activation), else three bf16 pieces, which have fp32's range; data gradient: f16 when the caller knows max |d_y| (a.amax),
  // else bf16.  (LeakyReLU(InstanceNorm(.)) is bounded by sqrt(V): beyond 2^24 voxels x 2^4 could overflow f16.)
  const bool f16 = X3_F16_FWD && (int64_t)a.D * a.H * a.W < (1ll << 24) && (mode == 0 ? (!x_free || a.amax != nullptr) : a.amax != nullptr);
  if (!f16) a.amax = nullptr;
  a.wpk = x3_weights(step, w, ws, a.Cin, a.Cout, mode, f16 ? 2 : 3, p, s);
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.nchunk = p.nchunk; a.ZC = p.zc; a.nitems = p.nitems;
  const dim3 grid(((p.nitems + 7) / 8) * 8);
#define X3_L(CIN_, P_, TY_, WL_, NPC_) hipLaunchKernelGGL((conv_x3_kernel<CIN_, P_, TY_, WL_, NORM, STATS, NPC_>), grid, dim3(NTHR), 0, s, a)
#define X3_D(NPC_) do { \
    if (p.cin_t == 4) { if (p.P == 2) X3_L(4, 2, 16, false, NPC_); else X3_L(4, 1, 8, false, NPC_); } \
    else if (p.cin_t == 8) { if (p.P == 2) X3_L(8, 2, 16, false, NPC_); else X3_L(8, 1, 8, false, NPC_); } \
    else X3_L(16, 1, 16, false, NPC_); } while (0)
  if (f16) X3_D(2); else X3_D(3);
#undef X3_D
#undef X3_L
  return modet_launch_status();
}

// data gradient + InstanceNorm-backward statistics of its output (STATS = 2)
int x3_launch_bst(modet_step_ctx* step, const X3Args& a0, const float* w, void* ws, int B, int mode, const X3Plan& p, hipStream_t s) {
  X3Args a = a0;
  const bool f16 = X3_F16_FWD && a.amax != nullptr && (int64_t)a.D * a.H * a.W < (1ll << 24);
  if (!f16) a.amax = nullptr;
  a.wpk = x3_weights(step, w, ws, a.Cin, a.Cout, mode, f16 ? 2 : 3, p, s);
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.nchunk = p.nchunk; a.ZC = p.zc; a.nitems = p.nitems;
  const dim3 grid(((p.nitems + 7) / 8) * 8);
#define X3_L(CIN_, P_, TY_, NPC_) hipLaunchKernelGGL((conv_x3_kernel<CIN_, P_, TY_, false, false, 2, NPC_>), grid, dim3(NTHR), 0, s, a)
#define X3_D(NPC_) do { \
    if (p.cin_t == 4) { if (p.P == 2) X3_L(4, 2, 16, NPC_); else X3_L(4, 1, 8, NPC_); } \
    else if (p.cin_t == 8) { if (p.P == 2) X3_L(8, 2, 16, NPC_); else X3_L(8, 1, 8, NPC_); } \
    else X3_L(16, 1, 16, NPC_); } while (0)
  if (f16) X3_D(2); else X3_D(3);
#undef X3_D
#undef X3_L
  return modet_launch_status();
}

// bf16 storage (one piece): x fp32 | bf16, y fp32 | bf16 (never both fp32: that is the fp32 path above)
template <bool IN16, bool OUT16, bool STATS>
int x3_launch16(modet_step_ctx* step, const X3Args& a0, const float* w, void* ws, int B, int mode, const X3Plan& p, hipStream_t s) {
  X3Args a = a0;
  a.wpk = x3_weights(step, w, ws, a.Cin, a.Cout, mode, 1, p, s);
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.nchunk = p.nchunk; a.ZC = p.zc; a.nitems = p.nitems;
  const dim3 grid(((p.nitems + 7) / 8) * 8);
#define X3_L(CIN_, P_) hipLaunchKernelGGL((conv_x3_kernel<CIN_, P_, 16, false, false, STATS, 1, IN16, OUT16>), grid, dim3(NTHR), 0, s, a)
  if (p.cin_t == 4) {
    if constexpr (!IN16) { if (p.P == 2) X3_L(4, 2); else X3_L(4, 1); } else return MODET_ERR_UNSUPPORTED;
  } else if (p.cin_t == 8) { if (p.P == 2) X3_L(8, 2); else X3_L(8, 1); }
  else { if (p.P == 2) X3_L(16, 2); else X3_L(16, 1); }
#undef X3_L
  return modet_launch_status();
}

}  // namespace

// ------------------------------------------------------------------------------------------------ weight gradient
// d_w[co][ci][tap] = sum_v d_y[v][co] * x[v + off(tap)][ci]  as  D[m][n] += A[m][k] B[k][n]  with  k = 32 voxels along W,
// m = (tap, ci) in tiles of 16 rows, n = cout -- the mapping of conv3d_bf16_wgrad_kernel (conv3d_bf16.hip): both operands need
// 8 CONSECUTIVE VOXELS of one channel per lane, so planes are staged channel-planar (two x-adjacent voxels per ds_write_b32),
// and the three x taps of a (dz, dy, ci) row come from one set of aligned reads by funnel shifts.  Here in fp32 accuracy
// (x and d_y split into three bf16 pieces each, six piece products per tile) and as a z-MARCH: a workgroup owns a 4 x 32
// column (y, x) and walks a chunk of planes; every x plane (one-voxel halo in y and x) is loaded, split and transposed
// ONCE and stays in a 4-slot LDS ring while the three d_y planes that touch it pass by, d_y planes are double-buffered.
// Workgroups are persistent over (column, z chunk) items and keep their fp32 accumulators (all 27 taps x Cin x 16 couts +
// the bias row) in registers across all of them; the four waves are summed through LDS at the end and ONE partial per
// workgroup goes to the fixed-order fp64 reduction shared with the bf16 path (deterministic, no atomics).
constexpr int WTX = 32, WHXP = 48, WTY = 4, WHY = WTY + 2;
struct X3WArgs {
  const void* x; const void* dy; float* part;                    // fp32; bf16 in the one-piece (storage) form
  int D, H, W, Cin, Cout, tiles_x, tiles_y, nchunk, ZC, nitems;
  const float* amax;                                             // NPC 2: amax[0] >= max |d_y| (device memory)
  const float* xamax;                                            // NPC 2: maxima of |x| (device memory) or null = x is an activation (fixed scale)
  const float* in_mean; const float* in_rstd;                    // NORM: x is a RAW ConvInsBlock output, LeakyReLU((x - mean) * rstd) while staged
};

// NP (Cout <= 8, "N-packed"): the 16 N columns are (q in {0,1}) x 8 couts, column block q = 1 multiplies d_y shifted one
// voxel in +x; with A row tiles t in {0,1} = x shifted by t voxels, D[t][q] = sum_u x[u + t - q] d_y[u] is the weight
// gradient of tap dx = t - q: (0,1), (0,0), (1,0) are dx = -1, 0, +1 and (1,1) is not a tap -- two row tiles per group
// instead of three, a third fewer MFMAs.  The q = 1 columns see the segment shifted by +1: what they miss over a whole row
// (u = 0, tap dx = -1) multiplies the zero padding x[-1], so nothing is lost; d_y rows carry one extra voxel.
// NPC = 1 (BASELINE.json configs[4], bf16 storage): d_y is bf16 in HBM, x is bf16 (X16) or fp32 rounded while staged; one
// piece per operand, one MFMA per tile instead of six, no split -- conv3d_bf16_wgrad_kernel's contract in this structure.
// NPC = 2 (round 5): fp32 accuracy on TWO f16 pieces per operand, three products per tile (see split2_h above).  x is an
// activation (scaled by 2^4; 4 channels: the un-normalised output of ConvBlock 1 -> 4, unscaled, like the forward launch), d_y a
// gradient whose maximum the caller hands over (a.amax, left by the InstanceNorm backward that produced d_y): scaled by the power
// of two that takes it to [2^14, 2^15).  The partial tiles are scaled back (exact) before they leave the workgroup.
// NORM (round 5): x is the raw output of the previous ConvInsBlock and is normalised while it is staged (zero padding stays
// zero) -- the normalised tensor of a ConvInsBlock -> ConvInsBlock chain then never exists in HBM in training either (the forward
// conv has had this form since round 2: conv_x3_kernel NORM); 5 VALU operations per staged element on the x operand.
template <int CIB, int NCO, bool NP, int NPC = 3, bool X16 = false, bool NORM = false>
__global__ __launch_bounds__(NTHR, NPC == 3 ? 2 : 3) void conv_x3_wgrad_kernel(const X3WArgs a) {
  static_assert(!NORM || (NPC != 1 && !X16), "the lazily normalised input belongs to the fp32 forms");
  static_assert(!NP || NCO == 8, "N packing: 2 x 8 couts");
  static_assert(NPC == 3 || NPC == 2 || NPC == 1, "three bf16 / two f16 pieces (fp32 accuracy) or one (bf16 storage)");
  static_assert(NPC == 1 || !X16, "bf16 x belongs to the one-piece form");
  constexpr bool D16 = NPC == 1;                                 // d_y is bf16 in HBM
  constexpr int U = CIB == 8 ? 5 : 3;                            // tap groups: 2 (dz,dy) combos x 8 channels, or 4 x 4
  constexpr int NT = NP ? 2 : 3;                                 // row tiles per group
  constexpr int PX = WHY * WHXP + 8;                             // elements of one channel's plane (+8: spreads the planes over the banks)
  constexpr int XPIECE = CIB * PX, XSLOT = NPC * XPIECE;
  constexpr int DROW = NP ? 40 : WTX;                            // d_y row: 32 voxels (+1 for the shifted columns, padded to 16 bytes)
  constexpr int DPAIRS = NP ? 17 : 16;
  constexpr int PD = WTY * DROW + 8, DPIECE = NCO * PD, DSLOT = NPC * DPIECE;
  constexpr int RED_FL = (U * NT + 1) * 256;
  constexpr int LDS_EL = (4 * XSLOT + 2 * DSLOT) * 2 >= RED_FL * 4 ? 4 * XSLOT + 2 * DSLOT : RED_FL * 2;   // (the cross-wave reduction re-uses the plane buffers)
  __shared__ __attribute__((aligned(16))) unsigned short lds[LDS_EL];
  unsigned short* xs = lds;
  unsigned short* dys = lds + 4 * XSLOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;

  // per-lane A row of each group: channel and (dz, dy) combination
  const int myci = CIB == 8 ? (li & 7) : (li & 3);
  int adz[U], aoff[U];                                           // dz of the row's combination; element offset inside an x slot (piece 0)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int combo = CIB == 8 ? 2 * u + (li >> 3) : 4 * u + (li >> 2);
    if (combo > 8) combo = 8;                                    // dummy rows: finite data, never written out
    adz[u] = combo / 3;
    aoff[u] = myci * PX + (combo % 3) * WHXP + 8 * lk;
  }
  const int bco = NP ? (li & 7) : (li < NCO ? li : li - NCO);    // N columns past the staged couts repeat valid planes (never read back)
  const bool bq = NP && (li >> 3);                               // this lane's column takes d_y shifted by one voxel
  const int boff = bco * PD + 8 * lk;

#ifdef MODET_TUNING
  long long dsum[6] = {0, 0, 0, 0, 0, 0};
#endif
  f32x4 acc[U][NT], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int d = 0; d < NT; ++d) acc[u][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned one2 = li == 0 ? (NPC == 2 ? 0x3c003c00u : 0x3f803f80u) : 0u;   // bf16 / f16 (1, 1): row 0 of the bias tile
  float xsc = 1.f, dsc = 1.f, dinv = 1.f;                        // NPC 2: operand scales (powers of two)
  if constexpr (NPC == 2) {
    xsc = CIB == 4 ? 1.f : X3_F16_XSCALE;
    if (a.xamax) { float xinv; x3_dyn_scale(a.xamax, xsc, xinv); }
    x3_dyn_scale(a.amax, dsc, dinv);
  }
  const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(one2, one2, one2, one2));

  // ---- staging maps (constant over the march): x item = (halo row hy, voxel pair, 4-channel group), d_y item = (row, pair, group)
  constexpr int QX = CIB / 4, NXI = WHY * 18 * QX, QD = NCO / 4, NDI = WTY * DPAIRS * QD;
  static_assert(NXI <= NTHR && NDI <= NTHR, "one staging item of each kind per thread");
  const bool xon = tid < NXI, don = tid < NDI;
  const int xc4 = tid % QX, xpr = (tid / QX) % 18, xhy = tid / (QX * 18);
  const int dc4 = tid % QD, dpr = (tid / QD) % DPAIRS, drow = tid / (QD * DPAIRS);
  const int xl = (xc4 * 4) * PX + xhy * WHXP + 2 * (xpr + 3);   // element of channel xc4*4, piece 0, inside an x slot
  const int dl = (dc4 * 4) * PD + drow * DROW + 2 * dpr;
  constexpr int XSZ = X16 ? 2 : 4, DSZ = D16 ? 2 : 4;            // bytes per element in HBM
  const unsigned in_plane_bytes = (unsigned)H * W * Cin * XSZ, dy_plane_bytes = (unsigned)H * W * Cout * DSZ;

  unsigned gx0 = X3_OOB, gx1 = X3_OOB, gd0 = X3_OOB, gd1 = X3_OOB;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
  const unsigned char* db_ = reinterpret_cast<const unsigned char*>(a.dy);
  struct Pair { float4 v0, v1; bool live; };                     // the two voxels of a thread's x pair / d_y pair (4 channels each;
                                                                 // bf16 tensors: 8 bytes per voxel in .x, .y); live: the plane exists
  float4 nm = make_float4(0.f, 0.f, 0.f, 0.f), nr = make_float4(1.f, 1.f, 1.f, 1.f);   // NORM: this thread's 4 channels of the item's sample
  auto ld4 = [&](const BufRsrc rs, unsigned off, bool is16) -> float4 {
    if (is16) {
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
      return make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), 0.f, 0.f);
    }
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  auto load_x = [&](int z) -> Pair {
    const bool live = z >= 0 && z < D;
    const BufRsrc rs = plane_rsrc(reinterpret_cast<const float*>(xb + (int64_t)(live ? z : 0) * in_plane_bytes), live ? in_plane_bytes : 0u);
    Pair p;
    p.v0 = ld4(rs, gx0, X16);
    p.v1 = ld4(rs, gx1, X16);
    p.live = live;
    return p;
  };
  auto load_dy = [&](int z, int ze) -> Pair {
    const bool live = z < ze;
    const BufRsrc rs = plane_rsrc(reinterpret_cast<const float*>(db_ + (int64_t)(live ? z : 0) * dy_plane_bytes), live ? dy_plane_bytes : 0u);
    Pair p;
    p.v0 = ld4(rs, gd0, D16);
    p.v1 = ld4(rs, gd1, D16);
    p.live = live;
    return p;
  };
  // a voxel pair's 4 channels -> packed (voxel, voxel + 1) words in the channel planes: split into three pieces each
  // (fp32 accuracy), or one bf16 piece (rounded here if the tensor is fp32, re-paired if it already is bf16)
  auto put = [&](unsigned short* base, int piece_el, int plane_el, const Pair& pr, bool is16, float sc) {
    if constexpr (NPC == 2) {
      const float p0[4] = {pr.v0.x * sc, pr.v0.y * sc, pr.v0.z * sc, pr.v0.w * sc}, p1[4] = {pr.v1.x * sc, pr.v1.y * sc, pr.v1.z * sc, pr.v1.w * sc};
      unsigned h[4], l[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) h[c] = pk_f16(p0[c], p1[c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a0, a1;
        unpk_f16(h[c], a0, a1);
        l[c] = pk_f16(p0[c] - a0, p1[c] - a1);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned* d = reinterpret_cast<unsigned*>(base + c * plane_el);
        d[0] = h[c]; d[piece_el / 2] = l[c];
      }
    } else if constexpr (NPC == 1) {
      unsigned w[4];
      if (is16) {
        const unsigned a0 = __float_as_uint(pr.v0.x), a1 = __float_as_uint(pr.v0.y);     // voxel 0: (c0 | c1 << 16), (c2 | c3 << 16)
        const unsigned b0 = __float_as_uint(pr.v1.x), b1 = __float_as_uint(pr.v1.y);     // voxel 1
        w[0] = __builtin_amdgcn_perm(b0, a0, 0x05040100u);      // (a0.lo, b0.lo)
        w[1] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);      // (a0.hi, b0.hi)
        w[2] = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
        w[3] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
      } else {
        w[0] = pk_bf16(pr.v0.x, pr.v1.x); w[1] = pk_bf16(pr.v0.y, pr.v1.y);
        w[2] = pk_bf16(pr.v0.z, pr.v1.z); w[3] = pk_bf16(pr.v0.w, pr.v1.w);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<unsigned*>(base + c * plane_el) = w[c];
    } else {
      const float p0[4] = {pr.v0.x, pr.v0.y, pr.v0.z, pr.v0.w}, p1[4] = {pr.v1.x, pr.v1.y, pr.v1.z, pr.v1.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h, m, l;
        split3_pk(p0[c], p1[c], h, m, l);
        unsigned* d = reinterpret_cast<unsigned*>(base + c * plane_el);
        d[0] = h; d[piece_el / 2] = m; d[piece_el] = l;          // piece stride in 32-bit words: piece_el / 2
      }
    }
  };
  auto store_x = [&](int slot, const Pair& pr0) {
    if (!xon) return;
    if constexpr (NORM) {                                        // out-of-volume voxels (the padding) stay exactly zero
      Pair pr = pr0;
      const bool k0 = pr.live && gx0 != X3_OOB, k1 = pr.live && gx1 != X3_OOB;
      pr.v0.x = k0 ? lrelu((pr.v0.x - nm.x) * nr.x) : 0.f; pr.v0.y = k0 ? lrelu((pr.v0.y - nm.y) * nr.y) : 0.f;
      pr.v0.z = k0 ? lrelu((pr.v0.z - nm.z) * nr.z) : 0.f; pr.v0.w = k0 ? lrelu((pr.v0.w - nm.w) * nr.w) : 0.f;
      pr.v1.x = k1 ? lrelu((pr.v1.x - nm.x) * nr.x) : 0.f; pr.v1.y = k1 ? lrelu((pr.v1.y - nm.y) * nr.y) : 0.f;
      pr.v1.z = k1 ? lrelu((pr.v1.z - nm.z) * nr.z) : 0.f; pr.v1.w = k1 ? lrelu((pr.v1.w - nm.w) * nr.w) : 0.f;
      put(xs + slot * XSLOT + xl, XPIECE, PX, pr, X16, xsc);
    } else {
      put(xs + slot * XSLOT + xl, XPIECE, PX, pr0, X16, xsc);
    }
  };
  auto store_dy = [&](int slot, const Pair& pr) { if (don) put(dys + slot * DSLOT + dl, DPIECE, PD, pr, D16, dsc); };

  // ---- one d_y plane (slot ds) against the three x planes around it: x plane of tap dz sits in ring slot (q + dz) & 3
  auto compute = [&](int q, int ds) {
    const int row = wave;                                        // WTY = 4 rows, one k-step per wave and plane
    const unsigned short* db = dys + ds * DSLOT + boff + row * DROW;
    bf16x8 b[NPC];
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
      const uint4 qq = *reinterpret_cast<const uint4*>(db + pc * DPIECE);
      if constexpr (NP) {
        const unsigned r0 = *reinterpret_cast<const unsigned*>(db + pc * DPIECE + 8);
        const unsigned s0 = __builtin_amdgcn_alignbit(qq.y, qq.x, 16), s1 = __builtin_amdgcn_alignbit(qq.z, qq.y, 16),
                       s2 = __builtin_amdgcn_alignbit(qq.w, qq.z, 16), s3 = __builtin_amdgcn_alignbit(r0, qq.w, 16);
        b[pc] = __builtin_bit_cast(bf16x8, make_uint4(bq ? s0 : qq.x, bq ? s1 : qq.y, bq ? s2 : qq.z, bq ? s3 : qq.w));
      } else {
        b[pc] = __builtin_bit_cast(bf16x8, qq);
      }
    }
#pragma unroll
    for (int pc = NPC - 1; pc >= 0; --pc) accb = x3_mma<NPC == 2>(ones, b[pc], accb);
    // raw operand words of a group (per piece: the aligned 16-byte block and its neighbours) are read one group AHEAD of
    // the MFMAs that use them, fenced: left alone the scheduler sinks each read to its first use and every group starts
    // with an exposed LDS round trip
    struct Raw { uint4 qq; unsigned r0, p3; };
    auto rd = [&](int u, Raw (&w)[NPC]) {
      const unsigned short* pa = xs + ((q + adz[u]) & 3) * XSLOT + aoff[u] + row * WHXP;
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        const unsigned short* pp = pa + pc * XPIECE;
        w[pc].qq = *reinterpret_cast<const uint4*>(pp + 8);
        w[pc].r0 = *reinterpret_cast<const unsigned*>(pp + 16);
        w[pc].p3 = NP ? 0u : *reinterpret_cast<const unsigned*>(pp + 6);
      }
    };
    Raw raw[2][NPC];
    rd(0, raw[0]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u + 1 < U) rd(u + 1, raw[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 f[NPC][NT];                                         // [piece][row tile]
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        const uint4 qq = raw[u & 1][pc].qq;
        const unsigned r0 = raw[u & 1][pc].r0;
        const unsigned a1 = __builtin_amdgcn_alignbit(qq.y, qq.x, 16), a2 = __builtin_amdgcn_alignbit(qq.z, qq.y, 16),
                       a3 = __builtin_amdgcn_alignbit(qq.w, qq.z, 16), a4 = __builtin_amdgcn_alignbit(r0, qq.w, 16);
        if constexpr (NP) {
          f[pc][0] = __builtin_bit_cast(bf16x8, qq);                              // x[v]
          f[pc][1] = __builtin_bit_cast(bf16x8, make_uint4(a1, a2, a3, a4));      // x[v + 1]
        } else {
          const unsigned a0 = __builtin_amdgcn_alignbit(qq.x, raw[u & 1][pc].p3, 16);
          f[pc][0] = __builtin_bit_cast(bf16x8, make_uint4(a0, a1, a2, a3));      // dx = 0: x[v - 1]
          f[pc][1] = __builtin_bit_cast(bf16x8, qq);
          f[pc][NT - 1] = __builtin_bit_cast(bf16x8, make_uint4(a1, a2, a3, a4));
        }
      }
      // six piece products, small terms first; the row-tile accumulators alternate (independent back-to-back MFMAs)
#define X3W(AP, BP)                                                                                      \
      _Pragma("unroll") for (int d = 0; d < NT; ++d)                                                      \
        acc[u][d] = x3_mma<NPC == 2>(f[AP][d], b[BP], acc[u][d]);
      if constexpr (NPC == 3) { X3W(2, 0) X3W(0, 2) X3W(1, 1) X3W(1, 0) X3W(0, 1) }
      if constexpr (NPC == 2) { X3W(1, 0) X3W(0, 1) }
      X3W(0, 0)
#undef X3W
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
    int t = item;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int zc = t % a.nchunk;
    const int b = t / a.nchunk;
    const int x0 = tx * WTX, y0 = ty * WTY, zs = zc * a.ZC;
    const int ze = zs + a.ZC < D ? zs + a.ZC : D;
    xb = reinterpret_cast<const unsigned char*>(a.x) + (int64_t)b * D * in_plane_bytes;
    db_ = reinterpret_cast<const unsigned char*>(a.dy) + (int64_t)b * D * dy_plane_bytes;
    {
      const int yy = y0 - 1 + xhy, xx = x0 - 8 + 2 * (xpr + 3);
      const bool rok = xon && yy >= 0 && yy < H && xc4 * 4 < Cin;
      gx0 = (rok && xx >= 0 && xx < W) ? (unsigned)(((yy * W + xx) * Cin + xc4 * 4) * XSZ) : X3_OOB;
      gx1 = (rok && xx + 1 >= 0 && xx + 1 < W) ? (unsigned)(((yy * W + xx + 1) * Cin + xc4 * 4) * XSZ) : X3_OOB;
      const int dyy = y0 + drow, dxx = x0 + 2 * dpr;
      const bool dok = don && dyy < H && dc4 * 4 < Cout;
      gd0 = (dok && dxx < W) ? (unsigned)(((dyy * W + dxx) * Cout + dc4 * 4) * DSZ) : X3_OOB;
      gd1 = (dok && dxx + 1 < W) ? (unsigned)(((dyy * W + dxx + 1) * Cout + dc4 * 4) * DSZ) : X3_OOB;
      if constexpr (NORM) {
        if (xon && xc4 * 4 < Cin) {
          nm = *reinterpret_cast<const float4*>(a.in_mean + b * Cin + xc4 * 4);
          nr = *reinterpret_cast<const float4*>(a.in_rstd + b * Cin + xc4 * 4);
        }
      }
    }
    __syncthreads();                                             // every wave is done with the previous item's planes
    // prologue: x planes zs-1, zs, zs+1 -> ring slots 0, 1, 2; d_y plane zs -> slot 0 (all four loads in flight together);
    // then two register sets: A <- (x plane zs+2, d_y plane zs+1), B <- (x plane zs+3, d_y plane zs+2).  A plane is loaded
    // TWO iterations before it is split: one iteration here is ~65 MFMAs per wave (a 4-row plane), shorter than an HBM
    // round trip under load, and the split would wait for its loads every plane (41 % of wave cycles parked, PMC)
    Pair xa, da, xb2, db2;
    {
      const Pair p0 = load_x(zs - 1), p1 = load_x(zs), p2 = load_x(zs + 1), pd = load_dy(zs, ze);
      xa = load_x(zs + 2); da = load_dy(zs + 1, ze);
      xb2 = load_x(zs + 3); db2 = load_dy(zs + 2, ze);
      store_x(0, p0); store_x(1, p1); store_x(2, p2); store_dy(0, pd);
    }
    __syncthreads();
    const int nz = ze - zs;
    // d_y plane zs + q with x planes zs + q - 1 .. zs + q + 1 in slots q .. q+2.  Two iterations per trip (register sets A, B);
    // an odd chunk runs one padded iteration: its d_y plane is past the chunk, loaded as zeros, and adds nothing
#ifdef MODET_TUNING
    long long tprev = clock64();
#endif
    for (int q = 0; q < nz; q += 2) {
      store_x((q + 3) & 3, xa);                                  // x plane zs+q+2 -> the slot x plane zs+q-2 left
      store_dy((q + 1) & 1, da);                                 // d_y plane zs+q+1 -> the other buffer
      X3_T(2)
      xa = load_x(zs + q + 4);
      da = load_dy(zs + q + 3, ze);
      X3_T(4)
      compute(q, q & 1);
      X3_T(0)
      __syncthreads();
      X3_T(3)
      store_x((q + 4) & 3, xb2);
      store_dy((q + 2) & 1, db2);
      X3_T(2)
      xb2 = load_x(zs + q + 5);
      db2 = load_dy(zs + q + 4, ze);
      X3_T(4)
      compute(q + 1, (q + 1) & 1);
      X3_T(0)
      __syncthreads();
      X3_T(3)
    }
  }
#ifdef MODET_TUNING
  if (g_x3_dbg && lane == 0) {
    long long* o = g_x3_dbg + ((int64_t)blockIdx.x * 4 + wave) * 6;
    for (int i = 0; i < 6; ++i) o[i] = dsum[i];
  }
#endif

  if constexpr (NPC == 2) {                                        // back from the f16 operands' scales (powers of two: exact)
    const float winv = dinv * (1.f / xsc);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int d = 0; d < NT; ++d) { acc[u][d][0] *= winv; acc[u][d][1] *= winv; acc[u][d][2] *= winv; acc[u][d][3] *= winv; }
    accb[0] *= dinv; accb[1] *= dinv; accb[2] *= dinv; accb[3] *= dinv;
  }
  // ---- sum the 4 waves through LDS (fixed order), one partial per workgroup
  float* red = reinterpret_cast<float*>(lds);
  for (int w4 = 0; w4 < 4; ++w4) {
    __syncthreads();
    if (wave == w4) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int d = 0; d < NT; ++d) {
          float4* slot = reinterpret_cast<float4*>(red + (u * NT + d) * 256) + lane;
          float4 v = make_float4(acc[u][d][0], acc[u][d][1], acc[u][d][2], acc[u][d][3]);
          if (w4 > 0) { const float4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *slot = v;
        }
      float4* slot = reinterpret_cast<float4*>(red + U * NT * 256) + lane;
      float4 v = make_float4(accb[0], accb[1], accb[2], accb[3]);
      if (w4 > 0) { const float4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *slot = v;
    }
  }
  __syncthreads();
  float* out = a.part + (size_t)blockIdx.x * RED_FL;
  for (int i = tid; i < RED_FL; i += NTHR) out[i] = red[i];
}

struct X3WPlan { int cib, nco, u, np, gx, tiles_x, tiles_y, nchunk, zc, nitems, red_fl; };
inline X3WPlan x3w_plan(int B, int D, int H, int W, int Cin, int Cout, int npc = 3) {
  X3WPlan p;
  p.cib = Cin <= 4 ? 4 : 8;
  p.nco = Cout <= 8 ? 8 : 16;
  p.u = p.cib == 8 ? 5 : 3;
  p.np = Cout <= 8;
  p.red_fl = (p.u * (p.np ? 2 : 3) + 1) * 256;
  p.tiles_x = cdiv(W, WTX);
  p.tiles_y = cdiv(H, WTY);
  const int cols = B * p.tiles_x * p.tiles_y;
  // resident workgroups (LDS): three bf16 pieces two per CU, one for 8 x 16; two f16 pieces (49 KB) three, two for x 16; one piece three
  const int slots = npc == 1 ? 768 : (npc == 2 ? (p.nco == 16 ? 512 : 768) : ((p.cib == 8 && p.nco == 16) ? 256 : 512));
  // chunks of >= 8 planes (3 planes of prologue each), about eight items per workgroup to even out the tail
  int n = (int)((8LL * slots + cols - 1) / cols);
  const int maxn = D / 8 > 0 ? D / 8 : 1;
  n = n < 1 ? 1 : (n > maxn ? maxn : n);
  p.zc = cdiv(D, n);
  p.nchunk = cdiv(D, p.zc);
  p.nitems = cols * p.nchunk;
  p.gx = p.nitems < slots ? p.nitems : slots;
  return p;
}

#ifdef MODET_TUNING
extern "C" int modet_debug_x3_timing(long long* buf) {       // not in the header: tuning builds only
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x3_dbg), &buf, sizeof(buf));
}
#endif

// ---- internal interface for conv3d.hip (C++ linkage, not part of the ABI): the fp32 entry points route eligible shapes here
bool modetx_x3_eligible(int B, int D, int H, int W, int Cin, int Cout) {
  // per-lane byte offsets inside one plane are 32-bit and 0x80000000 is the out-of-bounds sentinel: planes stay < 2 GiB
  return x3_shape_ok(Cin, Cout) && (int64_t)D * H * W >= 4096 && B <= 65535 && (int64_t)H * W * 16 * 4 < 0x7fffffffLL;
}
size_t modetx_x3_ws_bytes(int Cin, int Cout) { return x3_wpk_elems(16, 1) * sizeof(unsigned short); }
size_t modetx_x3_stats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const X3Plan p = x3_plan(B, D, H, W, Cin, Cout);
  return ((size_t)B * Cout + (size_t)p.nitems * Cout * 2) * sizeof(float);
}
// stats != null: stats = [B][Cout] shift header (filled by the caller's shift kernel) followed by the partial rows
int modetx_x3_conv(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats,
                   const float* in_mean, const float* in_rstd, int B, int D, int H, int W, int Cin, int Cout, int act, int mode,
                   hipStream_t s, const float* amax, bool x_free) {
  const X3Plan p = x3_plan(B, D, H, W, Cin, Cout);
  X3Args a{};
  a.amax = amax;             // data gradient: max |d_y|; forward: max |x| of an input whose range is only known on the device
  a.x = x; a.bias = bias; a.y = y; a.in_mean = in_mean; a.in_rstd = in_rstd;
  a.shift = stats; a.stats_rows = stats ? stats + (size_t)B * Cout : nullptr;
  a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act;
  if (in_mean) return stats ? x3_launch<true, true>(step, a, w, ws, B, mode, p, s, false) : x3_launch<true, false>(step, a, w, ws, B, mode, p, s, false);
  return stats ? x3_launch<false, true>(step, a, w, ws, B, mode, p, s, x_free) : x3_launch<false, false>(step, a, w, ws, B, mode, p, s, x_free);
}
// data gradient d_x = conv^T(d_y) (Cout channels in, Cin out) whose output is the gradient w.r.t. LeakyReLU(InstanceNorm(xraw)):
// also writes rows [B][items][Cin][2] of (sum g, sum g*xhat) -- the first pass of the InstanceNorm backward, for free
size_t modetx_x3_bst_rows_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const X3Plan p = x3_plan(B, D, H, W, Cout, Cin);
  return (size_t)p.nitems * Cin * 2 * sizeof(float);
}
int modetx_x3_dgrad_bst(modet_step_ctx* step, const float* dy, const float* w, float* dx, const float* xraw, const float* mean,
                        const float* rstd, float* rows, void* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t s,
                        const float* amax) {
  const X3Plan p = x3_plan(B, D, H, W, Cout, Cin);
  X3Args a{};
  a.amax = amax;
  a.x = dy; a.bias = nullptr; a.y = dx;
  a.xraw = xraw; a.bmean = mean; a.brstd = rstd; a.stats_rows = rows;
  a.D = D; a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin; a.act = 0;
  return x3_launch_bst(step, a, w, ws, B, 1, p, s);
}
// ---- bf16 storage (conv3d_bf16.hip routes the full-resolution layers here): x fp32 | bf16, y fp32 | bf16, one bf16 piece
bool modetx_x3_bf16_eligible(int B, int D, int H, int W, int Cin, int Cout, int x_bf16) {
  return x3_shape_ok(Cin, Cout) && (x_bf16 ? Cin % 8 == 0 : true) && (int64_t)D * H * W >= 4096 && B <= 65535 &&
         (int64_t)H * W * 16 * 4 < 0x7fffffffLL;
}
int modetx_x3_bf16_rows_per_sample(int B, int D, int H, int W, int Cin, int Cout) {
  const X3Plan p = x3_plan(B, D, H, W, Cin, Cout, 1);
  return p.nitems / B;
}
// stats != null: [B][Cout] shift header (filled by the caller) followed by one row [Cout][2] per (sample, workgroup item)
int modetx_x3_bf16_conv(modet_step_ctx* step, const void* x, int x_bf16, const float* w, const float* bias, void* y, int y_bf16,
                        void* ws, float* stats, int B, int D, int H, int W, int Cin, int Cout, int mode, hipStream_t s) {
  const X3Plan p = x3_plan(B, D, H, W, Cin, Cout, 1);
  X3Args a{};
  a.x = x; a.bias = bias; a.y = y;
  a.shift = stats; a.stats_rows = stats ? stats + (size_t)B * Cout : nullptr;
  a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = 0;
  if (!x_bf16 && !y_bf16) return MODET_ERR_UNSUPPORTED;
  if (!y_bf16) return stats ? MODET_ERR_UNSUPPORTED : x3_launch16<true, false, false>(step, a, w, ws, B, mode, p, s);
  if (x_bf16) return stats ? x3_launch16<true, true, true>(step, a, w, ws, B, mode, p, s) : x3_launch16<true, true, false>(step, a, w, ws, B, mode, p, s);
  return stats ? x3_launch16<false, true, true>(step, a, w, ws, B, mode, p, s) : x3_launch16<false, true, false>(step, a, w, ws, B, mode, p, s);
}
// ---- weight gradient
int modetx_wgrad_partials_reduce(modet_step_ctx* defer, const float* part, float* red, float* dw, float* db, int gx, int Cin,
                                 int Cout, int cib, int u, int layout, hipStream_t s);      // conv3d_bf16.hip
bool modetx_x3_wgrad_eligible(int B, int D, int H, int W, int Cin, int Cout) {
  return Cin % 4 == 0 && Cin >= 4 && Cin <= 8 && Cout % 4 == 0 && Cout >= 4 && Cout <= 16 && (int64_t)B * D * H * W >= 200000 &&
         (int64_t)H * W * 16 * 4 < 0x7fffffffLL;
}
size_t modetx_x3_wgrad_ws_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const X3WPlan p = x3w_plan(B, D, H, W, Cin, Cout);
  return ((size_t)768 + 1) * p.red_fl * sizeof(float);           // workgroup partials (up to 768 resident workgroups) + their column sums
}
int modetx_x3_wgrad(modet_step_ctx* defer, const float* x, const float* dy, float* dw, float* db, void* ws, int B, int D, int H,
                    int W, int Cin, int Cout, hipStream_t s, const float* amax, const float* in_mean, const float* in_rstd, const float* xamax) {
  const bool f16p = X3_F16_FWD && amax != nullptr && (int64_t)D * H * W < (1ll << 24);
  const X3WPlan p = x3w_plan(B, D, H, W, Cin, Cout, f16p ? 2 : 3);
  // two f16 pieces when the caller knows max |d_y| (and vouches for x: an activation), else three bf16 pieces
  const bool f16 = X3_F16_FWD && amax != nullptr && (int64_t)D * H * W < (1ll << 24);
  X3WArgs a{x, dy, (float*)ws, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.nchunk, p.zc, p.nitems, f16 ? amax : nullptr, f16 ? xamax : nullptr, in_mean, in_rstd};
#define X3W_D(NPC_, NORM_) do { \
    if (p.cib == 4) { \
      if (p.np) hipLaunchKernelGGL((conv_x3_wgrad_kernel<4, 8, true, NPC_, false, NORM_>), dim3(p.gx), dim3(NTHR), 0, s, a); \
      else hipLaunchKernelGGL((conv_x3_wgrad_kernel<4, 16, false, NPC_, false, NORM_>), dim3(p.gx), dim3(NTHR), 0, s, a); \
    } else { \
      if (p.np) hipLaunchKernelGGL((conv_x3_wgrad_kernel<8, 8, true, NPC_, false, NORM_>), dim3(p.gx), dim3(NTHR), 0, s, a); \
      else hipLaunchKernelGGL((conv_x3_wgrad_kernel<8, 16, false, NPC_, false, NORM_>), dim3(p.gx), dim3(NTHR), 0, s, a); \
    } } while (0)
  if (in_mean) { if (f16) X3W_D(2, true); else X3W_D(3, true); }
  else { if (f16) X3W_D(2, false); else X3W_D(3, false); }
#undef X3W_D
  float* red = (float*)ws + (size_t)p.gx * p.red_fl;
  return modetx_wgrad_partials_reduce(defer, (const float*)ws, red, dw, db, p.gx, Cin, Cout, p.cib, p.u, p.np ? 1 : 0, s);
}

// bf16 storage: x fp32 | bf16 (Cin 8), d_y bf16; same partial layout and reduction as the fp32 form
// Cin = 8 only: with one piece a plane of the 4 x 32 column is 6-10 MFMAs per wave, and for Cin = 4 the per-plane barrier
// and LDS round trip outweigh them (measured 4->8 at 160x192x224, B = 2: 0.327 ms against the tiled kernel's 0.261)
bool modetx_x3_bf16_wgrad_eligible(int B, int D, int H, int W, int Cin, int Cout, int x_bf16) {
  (void)x_bf16;
  return modetx_x3_wgrad_eligible(B, D, H, W, Cin, Cout) && Cin == 8 && Cout == 8;      // (8->16 at level 2: 0.143 vs 0.101 ms)
}
size_t modetx_x3_bf16_wgrad_ws_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const X3WPlan p = x3w_plan(B, D, H, W, Cin, Cout, 1);
  return ((size_t)768 + 1) * p.red_fl * sizeof(float);
}
int modetx_x3_bf16_wgrad(modet_step_ctx* defer, const void* x, int x_bf16, const void* dy, float* dw, float* db, void* ws, int B,
                         int D, int H, int W, int Cin, int Cout, hipStream_t s) {
  const X3WPlan p = x3w_plan(B, D, H, W, Cin, Cout, 1);
  X3WArgs a{x, dy, (float*)ws, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.nchunk, p.zc, p.nitems, nullptr, nullptr, nullptr};
#define X3W_L(CIB_, NCO_, NP_, X16_) hipLaunchKernelGGL((conv_x3_wgrad_kernel<CIB_, NCO_, NP_, 1, X16_>), dim3(p.gx), dim3(NTHR), 0, s, a)
  if (p.cib == 4) {
    if (x_bf16) return MODET_ERR_UNSUPPORTED;
    if (p.np) X3W_L(4, 8, true, false); else X3W_L(4, 16, false, false);
  } else if (x_bf16) {
    if (p.np) X3W_L(8, 8, true, true); else X3W_L(8, 16, false, true);
  } else {
    if (p.np) X3W_L(8, 8, true, false); else X3W_L(8, 16, false, false);
  }
#undef X3W_L
  float* red = (float*)ws + (size_t)p.gx * p.red_fl;
  return modetx_wgrad_partials_reduce(defer, (const float*)ws, red, dw, db, p.gx, Cin, Cout, p.cib, p.u, p.np ? 1 : 0, s);
}

void modetx_q_prepack_begin(modet_step_ctx* c, hipStream_t stream);      // conv3d_q.hip: layout 4
// the recorded 16-bit packing jobs with layout 2 / 3 belong to this file (conv3d_bf16.hip's prepack launch skips layouts >= 2)
void modetx_x3_prepack_begin(modet_step_ctx* c, hipStream_t stream) {
  std::vector<PackBKey> jobs;
  std::vector<size_t> off;
  unsigned short* arena;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    jobs = c->bjobs; off = c->boff; arena = c->barena;
  }
  X3PackTable t;
  t.n = 0;
  auto go = [&]() {
    if (t.n) hipLaunchKernelGGL(x3_pack_many_kernel, dim3(8, t.n), dim3(256), 0, stream, t);
    t.n = 0;
  };
  for (size_t i = 0; i < jobs.size(); ++i) {
    const PackBKey& k = jobs[i];
    if (k.layout != 2 && k.layout != 3) continue;        // (layout 4: conv3d_q.hip)
    t.job[t.n++] = X3PackJob{k.w, arena + off[i], k.Cin, k.Cout, k.CK, k.layout - 1, k.mode, k.npiece};
    if (t.n == X3PACK_MAX_JOBS) go();
  }
  go();
  modetx_q_prepack_begin(c, stream);
}
