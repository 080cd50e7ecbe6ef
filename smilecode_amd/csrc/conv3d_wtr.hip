// Weight gradient of the 3x3x3 / stride 1 / zero-pad 1 convolution for the MANY-CHANNEL and ODD-CHANNEL layers (encoder levels
// 2-5: 16->16 ... 128->128; the CWM layers 6/12/24/48 -> 12/24/48 and -> 2/4/8) in fp32 accuracy on gfx950's bf16 matrix
// pipe ("bf16x3", see conv3d_x3.hip), with the MFMA operands fetched by the LDS TRANSPOSE READ ds_read_b64_tr_b16.
//   reference call sites: the weight gradients autograd derives for nn.Conv3d in ConvInsBlock / CWM, ModeT/models.py:135-151,
//   :249-256 (arithmetic lives in ATen/MIOpen there).
//
//   d_w[co][ci][tap] = sum_v x[v + off(tap)][ci] * d_y[v][co]      D[m][n] += A[m][k] B[k][n],  k = voxels
//
// The contraction index is the VOXEL, but channels-last tensors hold consecutive CHANNELS of one voxel: both MFMA operands
// want "8 consecutive k of one row/column per lane", i.e. a transposed image.  The earlier kernels (conv3d.hip
// conv3d_wgrad_kernel: exact-f32 MFMA; conv3d_bf16.hip / conv3d_x3.hip: channel-planar LDS planes written voxel pair by
// voxel pair) pay for that transposition in the staging pass.  Here the LDS image is the tensor's own layout -- [voxel][16
// channels] bf16 rows of 32 bytes, one image per bf16 piece -- written with plain 8-byte stores, and ds_read_b64_tr_b16
// transposes on the way out: within a 16-lane group, lane 4j + c supplies the address of 4 bf16 (row j = which k, chunk c =
// which 4 channels) and lane t receives {row 0..3} of column t (chunk t >> 2, element t & 3) (probed on the chip:
// tools/micro/tr16_probe.hip).  Every lane supplies its OWN address, so a chunk may point at any (dz, dy, channel quad): the
// 16 rows of an M tile are 4 consecutive entries of the list q = (dz, dy) * NQ + quad, and channel counts 12, 24, 48 (NQ = 3
// quads per block) or 6 (2 quads) fill M tiles without padding (the exact-f32 kernel runs 12 -> 2 at 9.5 TFLOP/s).
//
// k-step = 32 voxels = 4 rows (lane group kg) x 8 x of the 2 x 8 x 8 voxel tile; a lane's 8 k values are 8 consecutive x.
// The three x taps of a (dz, dy) come from ONE set of reads: three 4-voxel blocks = slots x - 1 .. x + 10 of the halo'd row
// give the fragments of dx = 0 (registers 0..3), dx = 2 (registers 1..4) and dx = 1 (four v_alignbit) -- 9 transpose reads
// per 18 MFMAs (a first version read every tap separately: 1 read per MFMA, LDS-bound at 0.4 of the matrix pipe's rate).
// Row pitch 12 voxel slots (384 B) for x and an x ^ 4 (y & 1) swizzle for d_y put the two rows a 32-lane half reads on
// disjoint bank halves.  x3 arithmetic: operands hi + mid + lo (exact to 2^-24), six piece products of order <= 2, small
// terms first, fp32 accumulate; the three taps of a family alternate accumulators (no dependent MFMA chains).
//
// Work split.  Workgroup = (voxel tiles [persistent over gridDim.x], channel block of NQ quads, cout block of NT x 16).
// Wave w owns k-step w of every tile (NT = 2: N tile w & 1, k-steps 2 (w >> 1) and + 1) and ALL M tiles: 27 (+ bias) x 4
// accumulator registers; at the end of the workgroup's life the waves are summed through LDS in fixed order and ONE
// fragment-major partial goes to memory, which the two-stage fp64 reduction of conv3d_bf16.hip (layout 2) sums over the
// workgroups (deterministic).  The bias gradient is one more accumulator (A = ones).
#include "common.h"
#include "step_ctx.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;

constexpr int NTHR = 256;
constexpr int TZ = 2, TY = 8, TX = 8, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int HVOX = HZ * HY * HX, VOX = TZ * TY * TX;
constexpr int ROWB = 32;                               // bytes of one voxel slot of a 16-channel bf16 image
constexpr int XP = 12;                                 // slots per halo'd x row (10 used: pitch 384 B = odd multiple of 128 B)
constexpr int XPL = HZ * HY * XP * ROWB;               // one piece of the x tile
constexpr int DPL = VOX * ROWB;                        // one (piece, N tile) of the d_y tile (x swizzled by 4 (y & 1))
constexpr unsigned WTR_OOB = 0x80000000u;              // tensors are < 2 GiB (checked on the host): this offset reads 0

__device__ __forceinline__ BufRsrc tensor_rsrc(const float* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
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
// ---- two f16 pieces (round 5, see conv3d_x3.hip split2_h / conv_x3_wgrad_kernel NPC = 2): x = an activation, scaled by 2^4;
// d_y scaled by the power of two its producer's maximum gives (WtrArgs::amax); three piece products instead of six; the partial
// tiles are scaled back (exact) before they leave the workgroup
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr float WTR_F16_XSCALE = 16.f;
#ifndef WTR_F16
#define WTR_F16 1                                      // 0: three bf16 pieces everywhere (A/B builds)
#endif
__device__ __forceinline__ void split2h_pk(float a, float b, unsigned& hi, unsigned& lo) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const h16x2 t = __builtin_convertvector((f32x2){a, b}, h16x2);
  hi = __builtin_bit_cast(unsigned, t);
  const h16x2 r = __builtin_convertvector((f32x2){a - (float)t[0], b - (float)t[1]}, h16x2);
  lo = __builtin_bit_cast(unsigned, r);
}
template <bool F16>
__device__ __forceinline__ f32x4 wtr_mma(bf16x8 a, bf16x8 b, f32x4 c) {     // (F16: the 16-byte fragments hold f16)
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint2 tr_read(const unsigned char* lds, int off) {
  typedef __attribute__((address_space(3))) v4s* lptr;
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off)));
}
__device__ __forceinline__ bf16x8 frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

#ifdef MODET_TUNING
__device__ long long* g_wtr_dbg = nullptr;
#endif

struct WtrArgs {
  const float* x; const float* dy; float* part;
  int B, D, H, W, Cin, Cout, tiles_x, tiles_y, tiles_z, ntiles, n_coblk;
  const float* amax;                                   // NP 2: MODET_AMAX_SLOTS maxima of |d_y|
};
// One launch, SEVERAL layers (round 5): the weight gradients of pyramid levels 3-5 and of the CWM layers are 17 launches per
// train step of 20-50 us each, most of them far too small to fill 256 CUs (level 5: 24 workgroups) and every one paying its
// own ramp-up and tail.  Nothing waits for a weight gradient before the end of the backward pass, so the step context queues
// them (modetx_wtr_wgrad with a deferring context) and the flush launches all queued layers of one kernel variant as ONE grid:
// workgroup b belongs to the job whose [first, first + gx * gy) range holds it.  Per workgroup nothing changes (same tiles,
// same order, same partial row), so the results are bit-identical to the one-launch-per-layer form.
constexpr int WTR_MAX_JOBS = 8;
struct WtrTable {
  WtrArgs job[WTR_MAX_JOBS];
  int first[WTR_MAX_JOBS + 1], gx[WTR_MAX_JOBS], gy[WTR_MAX_JOBS];
  int n;
};

#ifndef WTR_VARIANT
#define WTR_VARIANT 8
#endif
// WTR_VARIANT bit 1: no register prefetch of the next tile's global loads; bit 2: no operand prefetch of the next family
template <int NQ, int NT, bool VEC, int NP = 3>
__global__ __launch_bounds__(NTHR, 2) void conv_wgrad_tr_kernel(const WtrTable t) {
  static_assert(NP == 3 || NP == 2, "three bf16 pieces or two f16 pieces");
  int job = 0;
  while (job + 1 < t.n && (int)blockIdx.x >= t.first[job + 1]) ++job;     // (scalar: blockIdx is uniform)
  const WtrArgs a = t.job[job];
  const int grid_x = t.gx[job], grid_y = t.gy[job];
  const int lb = (int)blockIdx.x - t.first[job];       // flat index inside the job's (grid_x, grid_y) grid, y fastest
  const int blk_x = lb / grid_y, blk_y = lb - blk_x * grid_y;
  constexpr int NF = (9 * NQ + 3) / 4;                 // families: 4 chunks of the list q = (dz, dy) * NQ + quad
  constexpr int MT = 3 * NF;                           // M tiles = (family, dx); slot MT = bias
  constexpr int KS = NT;                               // k-steps per wave and tile
  constexpr int XS_BYTES = NP * XPL, DS_BYTES = NP * NT * DPL;
  constexpr int RED_FL = (MT + 1) * NT * 256;
  constexpr int LDS_BYTES = XS_BYTES + DS_BYTES > RED_FL * 4 ? XS_BYTES + DS_BYTES : RED_FL * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

#ifdef MODET_TUNING
  const long long t_entry = __builtin_readcyclecounter();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, S = lane & 15, sj = S >> 2, sc = S & 3;
  const int nw = NT == 2 ? (wave & 1) : 0;             // this wave's N tile
  const int s_first = NT == 2 ? 2 * (wave >> 1) : wave;                 // its first k-step: z = s >> 1, rows 4 (s & 1) + kg
  const int cib = blk_y / a.n_coblk, cob = blk_y - cib * a.n_coblk;
  const int ci0 = cib * 4 * NQ, co0 = cob * 16 * NT;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;

  // supplier addresses: lane 4j + c of a group supplies slot x-block + j of chunk c
  int aoff[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    int q = 4 * f + sc;
    if (q >= 9 * NQ) q = 0;                            // dummy chunks: finite data, never reduced
    const int zy = q / NQ, quad = q - zy * NQ;
    aoff[f] = ((zy / 3) * HY + zy % 3) * XP * ROWB + quad * 8;
  }
  const int abase = (((s_first >> 1) * HY + 4 * (s_first & 1) + kg) * XP + sj) * ROWB;
  const int yrow = 4 * (s_first & 1) + kg;             // (NT = 2: the second k-step is 4 rows further, same parity)
  const int bslot = ((s_first >> 1) * TY + yrow) * TX;
  const int bsw = (kg & 1) << 2;
  const int boff0 = XS_BYTES + nw * DPL + (bslot + (sj ^ bsw)) * ROWB + sc * 8;           // x = 0..3
  const int boff1 = XS_BYTES + nw * DPL + (bslot + ((4 + sj) ^ bsw)) * ROWB + sc * 8;     // x = 4..7
  const unsigned one2 = S == 0 ? (NP == 2 ? 0x3c003c00u : 0x3f803f80u) : 0u;     // A = ones (hi piece): row 0 of the bias tile
  float dsc = 1.f, dinv = 1.f;                         // NP 2: scale of d_y and its inverse (powers of two)
  if constexpr (NP == 2) {
    float m = a.amax[lane * MODET_AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (m > 0.f && m < __builtin_huge_valf()) {
      int e;
      (void)frexpf(m, &e);
      int sh = 15 - e;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      dsc = ldexpf(1.f, sh); dinv = ldexpf(1.f, -sh);
    }
  }

  f32x4 acc[NF][3], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[f][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const BufRsrc rx = tensor_rsrc(a.x, (unsigned)((int64_t)a.B * D * H * W * Cin * 4));
  const BufRsrc rd = tensor_rsrc(a.dy, (unsigned)((int64_t)a.B * D * H * W * Cout * 4));

  // ---- staging roles (tile-invariant): the halo'd x tile is 40 rows (hz, hy) of IPR = 10 NQ 16-byte items; a pass covers RP
  // whole rows (RP | 10 or 10 | RP, so a pass's (hz, hy) are compile-time + per-thread constants: no divisions per item)
  constexpr int IPR = HX * NQ, RP = NQ >= 3 ? 5 : (NQ == 2 ? 10 : 20), NPASS = HZ * HY / RP;
  const int r_t = tid / IPR, i_t = tid - r_t * IPR;
  const bool x_act = r_t < RP;
  const int hx_t = i_t / NQ, qd_t = i_t - hx_t * NQ;
  const int hzo_t = RP == 20 ? r_t / HY : 0, hy_t = RP == 20 ? r_t - hzo_t * HY : r_t;
  const int cx_t = ci0 + qd_t * 4;
  const int xlds_t = ((hzo_t * HY + hy_t) * XP + hx_t) * ROWB + qd_t * 8;
  // d_y tile: 128 voxels x 4 NT items, a pass = 256 / (4 NT) voxels
  constexpr int DVP = NTHR / (4 * NT), NDP = VOX / DVP;
  const int dr_t = tid % (4 * NT), dv_t = tid / (4 * NT);
  const int cd_t = co0 + dr_t * 4;
  const int dlds_t = XS_BYTES + (dr_t >> 2) * DPL + (dr_t & 3) * 8;

  u32x4 xr[NPASS], dr[NDP];
  auto load_tile = [&](int tl) {
    int t = tl;
    const int x0 = (t % a.tiles_x) * TX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * TY; t /= a.tiles_y;
    const int z0 = (t % a.tiles_z) * TZ;
    const int vb = (t / a.tiles_z) * D;                // sample * D
    const int xx = x0 - 1 + hx_t;
    const bool xok = x_act && xx >= 0 && xx < W && cx_t < Cin;
    const unsigned xterm = ((unsigned)xx * (unsigned)Cin + (unsigned)cx_t) * 4u;
    const unsigned rowb = (unsigned)(W * Cin) * 4u;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row0 = i * RP;                         // compile-time: first row of the pass
      const int z = z0 - 1 + row0 / HY + hzo_t, yy = y0 - 1 + row0 % HY + hy_t;
      const bool ok = xok && z >= 0 && z < D && yy >= 0 && yy < H;
      const unsigned off = (unsigned)((vb + z) * H + yy) * rowb + xterm;
      if (VEC) {
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : WTR_OOB, 0, 0);
      } else {
        u32x4 v;
        v[0] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? off : WTR_OOB, 0, 0);
        v[1] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok && cx_t + 1 < Cin ? off + 4 : WTR_OOB, 0, 0);
        v[2] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok && cx_t + 2 < Cin ? off + 8 : WTR_OOB, 0, 0);
        v[3] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok && cx_t + 3 < Cin ? off + 12 : WTR_OOB, 0, 0);
        xr[i] = v;
      }
    }
    const unsigned rowd = (unsigned)(W * Cout) * 4u;
#pragma unroll
    for (int i = 0; i < NDP; ++i) {
      const int v = i * DVP + dv_t;                    // voxel of the tile: z = v >> 6, y = (v >> 3) & 7, x = v & 7
      const int xd = x0 + (v & 7), yy = y0 + ((v >> 3) & 7), z = z0 + (v >> 6);
      const bool ok = z < D && yy < H && xd < W && cd_t < Cout;
      const unsigned off = (unsigned)((vb + z) * H + yy) * rowd + ((unsigned)xd * (unsigned)Cout + (unsigned)cd_t) * 4u;
      if (VEC) {
        dr[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? off : WTR_OOB, 0, 0);
      } else {
        u32x4 w4;
        w4[0] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok ? off : WTR_OOB, 0, 0);
        w4[1] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok && cd_t + 1 < Cout ? off + 4 : WTR_OOB, 0, 0);
        w4[2] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok && cd_t + 2 < Cout ? off + 8 : WTR_OOB, 0, 0);
        w4[3] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok && cd_t + 3 < Cout ? off + 12 : WTR_OOB, 0, 0);
        dr[i] = w4;
      }
    }
  };
  auto write_tile = [&]() {
    if (x_act) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        unsigned char* dst = lds + xlds_t + i * RP * XP * ROWB;
        if constexpr (NP == 2) {
          unsigned h0, l0, h1, l1;
          split2h_pk(__uint_as_float(xr[i][0]) * WTR_F16_XSCALE, __uint_as_float(xr[i][1]) * WTR_F16_XSCALE, h0, l0);
          split2h_pk(__uint_as_float(xr[i][2]) * WTR_F16_XSCALE, __uint_as_float(xr[i][3]) * WTR_F16_XSCALE, h1, l1);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + XPL) = make_uint2(l0, l1);
        } else {
          unsigned h0, m0, l0, h1, m1, l1;
          split3_pk(__uint_as_float(xr[i][0]), __uint_as_float(xr[i][1]), h0, m0, l0);
          split3_pk(__uint_as_float(xr[i][2]), __uint_as_float(xr[i][3]), h1, m1, l1);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + XPL) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(dst + 2 * XPL) = make_uint2(l0, l1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NDP; ++i) {
      const int v = i * DVP + dv_t;
      const int slot = (v & ~7) | ((v & 7) ^ (((v >> 3) & 1) << 2));          // x ^ 4 (y & 1)
      unsigned char* dst = lds + dlds_t + slot * ROWB;
      if constexpr (NP == 2) {
        unsigned h0, l0, h1, l1;
        split2h_pk(__uint_as_float(dr[i][0]) * dsc, __uint_as_float(dr[i][1]) * dsc, h0, l0);
        split2h_pk(__uint_as_float(dr[i][2]) * dsc, __uint_as_float(dr[i][3]) * dsc, h1, l1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + NT * DPL) = make_uint2(l0, l1);
      } else {
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pk(__uint_as_float(dr[i][0]), __uint_as_float(dr[i][1]), h0, m0, l0);
        split3_pk(__uint_as_float(dr[i][2]), __uint_as_float(dr[i][3]), h1, m1, l1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + NT * DPL) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(dst + 2 * NT * DPL) = make_uint2(l0, l1);
      }
    }
  };

#ifdef MODET_TUNING
  long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tq0, tq1;      // cycles in: load issue | barrier 1 | split + LDS write (incl. the wait for the loads) | barrier 2 | MFMA phase
#define TPH(k) do { tq1 = __builtin_readcyclecounter(); tph[k] += tq1 - tq0; tq0 = tq1; } while (0)
#else
#define TPH(k) do { } while (0)
#endif
  int tile = blk_x;
  if (!(WTR_VARIANT & 2) && tile < a.ntiles) load_tile(tile);
#ifdef MODET_TUNING
  tq0 = __builtin_readcyclecounter();
  tph[5] = tq0 - t_entry;
#endif
  for (; tile < a.ntiles; tile += grid_x) {
    if (WTR_VARIANT & 2) load_tile(tile);
    TPH(0);
    __syncthreads();                                   // every wave is done with the previous tile's images
    TPH(1);
    if (WTR_VARIANT & 8) __builtin_amdgcn_s_setprio(2);     // staging is VALU work beside the other workgroup's MFMA phase
    write_tile();
    TPH(2);
    __syncthreads();
    TPH(3);
    if (!(WTR_VARIANT & 2) && tile + grid_x < a.ntiles) load_tile(tile + grid_x);
    if (WTR_VARIANT & 8) __builtin_amdgcn_s_setprio(0);
    TPH(0);
    // ---- MFMA phase.  Unit = (k-step ks, family f): 9 transpose reads -> fragments of dx = 0, 1, 2 -> 18 MFMAs on three
    // accumulators; the reads of the next unit are issued before the MFMAs of this one (left alone, hipcc emits
    // read -> lgkmcnt(0) -> MFMA chains).
    auto load_raw = [&](int u, uint2 (&R)[NP][3]) {
      const int ks = u / NF, f = u - ks * NF;
      const int o = abase + aoff[f] + ks * 4 * XP * ROWB;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int b = 0; b < 3; ++b) R[p][b] = tr_read(lds, o + p * XPL + b * 4 * ROWB);
    };
    uint2 Rc[NP][3], Rn[NP][3];
    bf16x8 bq[NP];
    load_raw(0, Rc);
#pragma unroll
    for (int u = 0; u < KS * NF; ++u) {
      const int ks = u / NF, f = u - ks * NF;
      if (f == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const uint2 lo = tr_read(lds, boff0 + p * NT * DPL + ks * 4 * TX * ROWB), hi = tr_read(lds, boff1 + p * NT * DPL + ks * 4 * TX * ROWB);
          bq[p] = frag(lo.x, lo.y, hi.x, hi.y);
        }
      }
      if (!(WTR_VARIANT & 4) && u + 1 < KS * NF) load_raw(u + 1, Rn);
      __builtin_amdgcn_sched_barrier(0);
      if (f == 0) {                                    // d_bias: ones x d_y (the mid / lo pieces of "ones" are zero)
        const bf16x8 on = frag(one2, one2, one2, one2);
        if constexpr (NP == 3) accb = wtr_mma<false>(on, bq[NP - 1], accb);
        accb = wtr_mma<NP == 2>(on, bq[1], accb);
        accb = wtr_mma<NP == 2>(on, bq[0], accb);
      }
      bf16x8 d0[NP], d1[NP], d2[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned r0 = Rc[p][0].x, r1 = Rc[p][0].y, r2 = Rc[p][1].x, r3 = Rc[p][1].y, r4 = Rc[p][2].x;
        d0[p] = frag(r0, r1, r2, r3);
        d2[p] = frag(r1, r2, r3, r4);
        d1[p] = frag(__builtin_amdgcn_alignbit(r1, r0, 16), __builtin_amdgcn_alignbit(r2, r1, 16),
                     __builtin_amdgcn_alignbit(r3, r2, 16), __builtin_amdgcn_alignbit(r4, r3, 16));
      }
#define MM3(PA, PB)                                                                                   \
      acc[f][0] = wtr_mma<NP == 2>(d0[PA], bq[PB], acc[f][0]);       \
      acc[f][1] = wtr_mma<NP == 2>(d1[PA], bq[PB], acc[f][1]);       \
      acc[f][2] = wtr_mma<NP == 2>(d2[PA], bq[PB], acc[f][2]);
      if constexpr (NP == 3) { MM3(NP - 1, 0) MM3(0, NP - 1) MM3(1, 1) }
      MM3(1, 0) MM3(0, 1) MM3(0, 0)
#undef MM3
      __builtin_amdgcn_sched_barrier(0);
      if (u + 1 < KS * NF) {
        if (WTR_VARIANT & 4) load_raw(u + 1, Rc);
        else {
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int b = 0; b < 3; ++b) Rc[p][b] = Rn[p][b];
        }
      }
    }
    TPH(4);
  }
  if constexpr (NP == 2) {                             // back from the f16 operands' scales (powers of two: exact)
    const float winv = dinv * (1.f / WTR_F16_XSCALE);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int d = 0; d < 3; ++d) { acc[f][d][0] *= winv; acc[f][d][1] *= winv; acc[f][d][2] *= winv; acc[f][d][3] *= winv; }
    accb[0] *= dinv; accb[1] *= dinv; accb[2] *= dinv; accb[3] *= dinv;
  }
  // ---- sum the waves that share an N tile through LDS in fixed order; ONE fragment-major partial per workgroup:
  // part[bx][by][mt <= MT][n][lane * 4 + reg]
  float* red = reinterpret_cast<float*>(lds);
  constexpr int ROUNDS = 4 / NT;
#pragma unroll 1
  for (int r = 0; r < ROUNDS; ++r) {
    __syncthreads();
    if ((NT == 2 ? (wave >> 1) : wave) == r) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float4* slot = reinterpret_cast<float4*>(red + ((f * 3 + d) * NT + nw) * 256) + lane;
          float4 v = make_float4(acc[f][d][0], acc[f][d][1], acc[f][d][2], acc[f][d][3]);
          if (r > 0) { const float4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *slot = v;
        }
      float4* slot = reinterpret_cast<float4*>(red + (MT * NT + nw) * 256) + lane;
      float4 v = make_float4(accb[0], accb[1], accb[2], accb[3]);
      if (r > 0) { const float4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *slot = v;
    }
  }
  __syncthreads();
  float* out = a.part + (size_t)lb * (size_t)RED_FL;   // = (blk_x * grid_y + blk_y): the row order the reduction expects
  for (int i = tid * 4; i < RED_FL; i += NTHR * 4) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(red + i);
#ifdef MODET_TUNING
  TPH(6);
  if (g_wtr_dbg && lane == 0) {
    long long* o = g_wtr_dbg + ((int64_t)(blk_y * grid_x + blk_x) * 4 + wave) * 8;
    for (int i = 0; i < 7; ++i) o[i] = tph[i];
    o[7] = t_entry;
  }
#endif
}

struct WtrPlan { int nq, nt, n_cib, n_coblk, gy, gx, tiles_x, tiles_y, tiles_z, ntiles, mt, red_fl; };
inline WtrPlan wtr_plan(int B, int D, int H, int W, int Cin, int Cout) {
  WtrPlan p;
  const int quads = (Cin + 3) / 4;
  p.nq = quads % 4 == 0 ? 4 : (quads % 3 == 0 ? 3 : (quads % 2 == 0 ? 2 : 1));
  p.n_cib = quads / p.nq;
  p.nt = Cout > 16 ? 2 : 1;
  p.n_coblk = cdiv(Cout, 16 * p.nt);
  p.gy = p.n_cib * p.n_coblk;
  p.tiles_x = cdiv(W, TX); p.tiles_y = cdiv(H, TY); p.tiles_z = cdiv(D, TZ);
  p.ntiles = B * p.tiles_x * p.tiles_y * p.tiles_z;
  int slots = 512;                                     // resident workgroups (LDS 58 / 71 KB, <= 256 registers: two per CU)
  if (const char e = modet_tuning_env("MODET_WTR_SLOTS")) slots = e == '1' ? 256 : (e == '2' ? 384 : 512);
  int gx = slots / p.gy;
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  p.gx = gx;
  p.mt = 3 * ((9 * p.nq + 3) / 4);
  p.red_fl = (p.mt + 1) * p.nt * 256;
  return p;
}

}  // namespace

#ifdef MODET_TUNING
extern "C" int modet_debug_wtr_timing(long long* buf) {       // not in the header: tuning builds only
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wtr_dbg), &buf, sizeof(buf));
}
#endif

// ---- internal interface for conv3d.hip (C++ linkage, not part of the ABI)
int modetx_wgrad_partials_reduce2(modet_step_ctx* defer, const float* part, float* red, float* dw, float* db, int gx, int gy,
                                  int Cin, int Cout, int nq, int mt, int nt, int n_coblk, hipStream_t s);      // conv3d_bf16.hip
bool modetx_wtr_eligible(int B, int D, int H, int W, int Cin, int Cout) {
  const int64_t n = (int64_t)B * D * H * W;
  const bool vec = Cin % 4 == 0 && Cout % 4 == 0;
  return Cin >= 4 && n * (Cin > Cout ? Cin : Cout) * 4 < 0x7fffffffLL && Cout <= 256 && Cin <= 1024 && (vec || Cout <= 16);
}
size_t modetx_wtr_ws_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const WtrPlan p = wtr_plan(B, D, H, W, Cin, Cout);
  const int gx = p.gy >= 512 ? 1 : 512 / p.gy;         // upper bound of the plan's gx
  return ((size_t)gx + 1) * p.gy * p.red_fl * sizeof(float);      // workgroup partials + their column sums
}
// launch the jobs of ONE kernel variant (nq, nt, vec) as one grid
static void wtr_launch(const WtrTable& t, int nq, int nt, bool vec, bool f16, hipStream_t s) {
  const dim3 grid(t.first[t.n]);
#define WTR_L(NQ_, NT_, V_, NP_) hipLaunchKernelGGL((conv_wgrad_tr_kernel<NQ_, NT_, V_, NP_>), grid, dim3(NTHR), 0, s, t)
#define WTR_Q(NT_, V_, NP_) do { if (nq == 4) WTR_L(4, NT_, V_, NP_); else if (nq == 3) WTR_L(3, NT_, V_, NP_); else if (nq == 2) WTR_L(2, NT_, V_, NP_); else WTR_L(1, NT_, V_, NP_); } while (0)
#define WTR_P(NP_) do { \
    if (nt == 2) WTR_Q(2, true, NP_);                  /* (eligibility: the odd channel counts only come with Cout <= 16) */ \
    else if (vec) WTR_Q(1, true, NP_); \
    else WTR_Q(1, false, NP_); } while (0)
  if (f16) WTR_P(2); else WTR_P(3);
#undef WTR_P
#undef WTR_Q
#undef WTR_L
}

// does a DEFERRED call keep reading x / d_y until the context's flush?  (the caller must keep them alive then)
#ifndef WTR_BATCH_MAX_VOXELS
#define WTR_BATCH_MAX_VOXELS 1500000      // (tools: build with -DWTR_BATCH_MAX_VOXELS=0 for the one-launch-per-layer A side)
#endif
bool modetx_wtr_batches(int B, int D, int H, int W) { return (int64_t)B * D * H * W <= WTR_BATCH_MAX_VOXELS; }

int modetx_wtr_wgrad(modet_step_ctx* defer, const float* x, const float* dy, float* dw, float* db, void* ws, int B, int D, int H,
                     int W, int Cin, int Cout, hipStream_t s, const float* amax) {
  const WtrPlan p = wtr_plan(B, D, H, W, Cin, Cout);
  const bool vec = Cin % 4 == 0 && Cout % 4 == 0;
  if (!WTR_F16 || (int64_t)D * H * W >= (1ll << 24)) amax = nullptr;     // (two f16 pieces: the caller knows max |d_y|, x is an activation)
  float* red = (float*)ws + (size_t)p.gx * p.gy * p.red_fl;
  if (defer && modetx_wtr_batches(B, D, H, W)) {       // queued: the flush launches it together with its variant's other layers
    WtrQueued j{x, dy, (float*)ws, B, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, p.ntiles, p.n_coblk, p.nq, p.nt, vec ? 1 : 0,
                p.gx, p.gy, amax};
    {
      std::lock_guard<std::mutex> lk(defer->mu);
      defer->wtrjobs.push_back(j);
    }
    return modetx_wgrad_partials_reduce2(defer, (const float*)ws, red, dw, db, p.gx, p.gy, Cin, Cout, p.nq, p.mt, p.nt, p.n_coblk, s);
  }
  WtrTable t;
  t.job[0] = WtrArgs{x, dy, (float*)ws, B, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, p.ntiles, p.n_coblk, amax};
  t.first[0] = 0; t.first[1] = p.gx * p.gy; t.gx[0] = p.gx; t.gy[0] = p.gy; t.n = 1;
  wtr_launch(t, p.nq, p.nt, vec, amax != nullptr, s);
  return modetx_wgrad_partials_reduce2(defer, (const float*)ws, red, dw, db, p.gx, p.gy, Cin, Cout, p.nq, p.mt, p.nt, p.n_coblk, s);
}

// the queued partial-tile launches of a deferring context, grouped by kernel variant (called by the flush BEFORE the reductions)
void modetx_wtr_flush(modet_step_ctx* c, hipStream_t s) {
  std::vector<WtrQueued> jobs;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    jobs.swap(c->wtrjobs);
  }
  std::vector<char> done(jobs.size(), 0);
  for (size_t i = 0; i < jobs.size(); ++i) {
    if (done[i]) continue;
    WtrTable t;
    t.n = 0; t.first[0] = 0;
    for (size_t k = i; k < jobs.size() && t.n < WTR_MAX_JOBS; ++k) {
      const WtrQueued& q = jobs[k];
      if (done[k] || q.nq != jobs[i].nq || q.nt != jobs[i].nt || q.vec != jobs[i].vec || (q.amax != nullptr) != (jobs[i].amax != nullptr)) continue;
      t.job[t.n] = WtrArgs{q.x, q.dy, q.part, q.B, q.D, q.H, q.W, q.Cin, q.Cout, q.tiles_x, q.tiles_y, q.tiles_z, q.ntiles, q.n_coblk, q.amax};
      t.gx[t.n] = q.gx; t.gy[t.n] = q.gy;
      t.first[t.n + 1] = t.first[t.n] + q.gx * q.gy;
      ++t.n;
      done[k] = 1;
    }
    for (int k = t.n; k < WTR_MAX_JOBS; ++k) { t.first[k + 1] = t.first[t.n]; t.gx[k] = 1; t.gy[k] = 1; }
    wtr_launch(t, jobs[i].nq, jobs[i].nt, jobs[i].vec != 0, jobs[i].amax != nullptr, s);
  }
}
