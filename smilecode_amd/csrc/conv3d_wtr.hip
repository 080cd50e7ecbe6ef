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
// voxel pair, x taps by funnel shifts) pay for that transposition in the staging pass.  Here the LDS image is the tensor's
// own layout -- [voxel][16 channels] bf16 rows of 32 bytes, one image per bf16 piece -- written with plain 8-byte stores,
// and ds_read_b64_tr_b16 transposes on the way out: within a 16-lane group, lane 4j + c supplies the address of 4 bf16
// (row j = which k, chunk c = which 4 channels) and lane t receives {row 0..3} of column t (chunk t >> 2, element t & 3)
// (probed on the chip: tools/micro/tr16_probe.hip).  Because every lane supplies its OWN address,
//   * a chunk may point at any (tap, channel quad): the 16 rows of an M tile are 4 consecutive entries of the list
//     q = tap * NQ + quad, so channel counts 12, 24, 48 (NQ = 3 quads per block) and 6 (2 quads) fill M tiles with no padding
//     (the exact-f32 kernel runs Cin = 12 as three 4-channel tiles and 12 -> 2 at 9.5 TFLOP/s);
//   * a tap is only an address offset: no shifted copies, no alignbit.
// k-step = 32 voxels = 4 (y) x 8 (x) of the 2 x 8 x 8 voxel tile: read r in {0,1} of lane group kg covers row 4 yq + 2 r +
// (kg >> 1), x = 4 (kg & 1) + j, so a 32-lane half reads 8 consecutive voxels = 256 contiguous bytes (conflict-free), and A
// and B use the same k order.  x3 arithmetic: operands hi + mid + lo (exact to 2^-24), six piece products of order <= 2,
// small terms first, fp32 accumulate.
//
// Work split.  Workgroup = (voxel tiles [persistent over gridDim.x], channel block of NQ quads, cout block of NT x 16).
// Its four waves own DISJOINT M tiles (mt = wave + 4 mi), so nothing is summed across waves: every wave stores its own
// accumulators as fragment-major partial tiles, which the two-stage fp64 reduction of conv3d_bf16.hip (layout 2) sums over
// the workgroups in fixed order (deterministic).  The bias gradient rides in the first free M-tile slot (A = ones).
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
constexpr int ROWB = 32;                               // bytes of one voxel row of a 16-channel bf16 image
constexpr int XPL = HVOX * ROWB, DPL = VOX * ROWB;     // one piece of the x tile / one (piece, N tile) of the d_y tile
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
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lds, int off0, int off1) {
  typedef __attribute__((address_space(3))) v4s* lptr;
  const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off0));
  const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct WtrArgs {
  const float* x; const float* dy; float* part;
  int B, D, H, W, Cin, Cout, tiles_x, tiles_y, tiles_z, ntiles, n_coblk;
};

template <int NQ, int NT>
#ifndef WTR_VARIANT
#define WTR_VARIANT 2
#endif
// WTR_VARIANT bit 0: two workgroups per CU for NT = 1 too (256 registers); bit 1: no register prefetch of the next tile
__global__ __launch_bounds__(NTHR, (NT == 1 && !(WTR_VARIANT & 1)) ? 3 : 2) void conv_wgrad_tr_kernel(const WtrArgs a) {
  constexpr int MT = (27 * NQ + 3) / 4;                // M tiles holding real (tap, quad) chunks; slot MT = bias tile
  constexpr int MT_W = (MT + 1 + 3) / 4;               // M-tile slots per wave
  constexpr int BIAS_W = MT % 4, BIAS_MI = MT / 4;
  constexpr int XS_BYTES = 3 * XPL, DS_BYTES = 3 * NT * DPL;
  constexpr int NXI = (HVOX * NQ + NTHR - 1) / NTHR;   // 16-byte x items per thread and tile
  constexpr int NDI = (VOX * NT * 4) / NTHR;           // 16-byte d_y items per thread and tile
  __shared__ __attribute__((aligned(16))) unsigned char lds[XS_BYTES + DS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, S = lane & 15, sj = S >> 2, sc = S & 3;
  const int xl = 4 * (kg & 1) + sj, yl = kg >> 1;
  const int cib = blockIdx.y / a.n_coblk, cob = blockIdx.y - cib * a.n_coblk;
  const int ci0 = cib * 4 * NQ, co0 = cob * 16 * NT;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;

  int aoff[MT_W];
#pragma unroll
  for (int mi = 0; mi < MT_W; ++mi) {
    int q = (wave + 4 * mi) * 4 + sc;
    if (q >= 27 * NQ) q = 0;                           // dummy chunks: finite data, never reduced
    const int tap = q / NQ, quad = q - tap * NQ;
    const int dz = tap / 9, dyy = (tap / 3) % 3, dx = tap % 3;
    aoff[mi] = ((dz * HY + dyy + yl) * HX + dx + xl) * ROWB + quad * 8;
  }
  const int boff = XS_BYTES + (yl * TX + xl) * ROWB + sc * 8;
  const bool bias_wave = wave == BIAS_W;
  const unsigned one2 = S == 0 ? 0x3f803f80u : 0u;     // A = ones: row 0 (lane & 15 == 0) of the bias tile, hi piece

  f32x4 acc[MT_W][NT];
#pragma unroll
  for (int mi = 0; mi < MT_W; ++mi)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[mi][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const BufRsrc rx = tensor_rsrc(a.x, (unsigned)((int64_t)a.B * D * H * W * Cin * 4));
  const BufRsrc rd = tensor_rsrc(a.dy, (unsigned)((int64_t)a.B * D * H * W * Cout * 4));
  const bool xvec = (Cin & 3) == 0, dvec = (Cout & 3) == 0;

  u32x4 xr[NXI], dr[NDI];
  auto load_tile = [&](int tl) {
    int t = tl;
    const int x0 = (t % a.tiles_x) * TX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * TY; t /= a.tiles_y;
    const int z0 = (t % a.tiles_z) * TZ;
    const int vb = (t / a.tiles_z) * D;                // sample * D
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int idx = tid + i * NTHR;
      const int hv = idx / NQ, qd = idx - hv * NQ;
      const int hx = hv % HX, t2 = hv / HX;
      const int hy = t2 % HY, hz = t2 / HY;
      const int z = z0 + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
      const int c = ci0 + qd * 4;
      const bool ok = idx < HVOX * NQ && z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W && c < Cin;
      const unsigned off = ((unsigned)(((vb + z) * H + yy) * W + xx) * (unsigned)Cin + (unsigned)c) * 4u;
      if (xvec) {
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : WTR_OOB, 0, 0);
      } else {
        u32x4 v;
        v[0] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? off : WTR_OOB, 0, 0);
        v[1] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok && c + 1 < Cin ? off + 4 : WTR_OOB, 0, 0);
        v[2] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok && c + 2 < Cin ? off + 8 : WTR_OOB, 0, 0);
        v[3] = __builtin_amdgcn_raw_buffer_load_b32(rx, ok && c + 3 < Cin ? off + 12 : WTR_OOB, 0, 0);
        xr[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int idx = tid + i * NTHR;
      const int v = idx / (NT * 4), r = idx - v * (NT * 4);      // voxel of the tile, (N tile, quad)
      const int xx = x0 + (v & 7), yy = y0 + ((v >> 3) & 7), z = z0 + (v >> 6);
      const int c = co0 + r * 4;
      const bool ok = z < D && yy < H && xx < W && c < Cout;
      const unsigned off = ((unsigned)(((vb + z) * H + yy) * W + xx) * (unsigned)Cout + (unsigned)c) * 4u;
      if (dvec) {
        dr[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? off : WTR_OOB, 0, 0);
      } else {
        u32x4 w4;
        w4[0] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok ? off : WTR_OOB, 0, 0);
        w4[1] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok && c + 1 < Cout ? off + 4 : WTR_OOB, 0, 0);
        w4[2] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok && c + 2 < Cout ? off + 8 : WTR_OOB, 0, 0);
        w4[3] = __builtin_amdgcn_raw_buffer_load_b32(rd, ok && c + 3 < Cout ? off + 12 : WTR_OOB, 0, 0);
        dr[i] = w4;
      }
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < HVOX * NQ) {
        const int hv = idx / NQ, qd = idx - hv * NQ;
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pk(__uint_as_float(xr[i][0]), __uint_as_float(xr[i][1]), h0, m0, l0);
        split3_pk(__uint_as_float(xr[i][2]), __uint_as_float(xr[i][3]), h1, m1, l1);
        unsigned char* dst = lds + hv * ROWB + qd * 8;
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + XPL) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(dst + 2 * XPL) = make_uint2(l0, l1);
      }
    }
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int idx = tid + i * NTHR;
      const int v = idx / (NT * 4), r = idx - v * (NT * 4);
      const int n = r >> 2, qd = r & 3;
      unsigned h0, m0, l0, h1, m1, l1;
      split3_pk(__uint_as_float(dr[i][0]), __uint_as_float(dr[i][1]), h0, m0, l0);
      split3_pk(__uint_as_float(dr[i][2]), __uint_as_float(dr[i][3]), h1, m1, l1);
      unsigned char* dst = lds + XS_BYTES + n * DPL + v * ROWB + qd * 8;
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(dst + NT * DPL) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(dst + 2 * NT * DPL) = make_uint2(l0, l1);
    }
  };

  int tile = blockIdx.x;
  if (!(WTR_VARIANT & 2) && tile < a.ntiles) load_tile(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    if (WTR_VARIANT & 2) load_tile(tile);
    __syncthreads();                                   // every wave is done with the previous tile's images
    write_tile();
    __syncthreads();
    if (!(WTR_VARIANT & 2) && tile + (int)gridDim.x < a.ntiles) load_tile(tile + gridDim.x);
    // ---- MFMA phase, software-pipelined by hand (left alone, hipcc emits read x 6 -> lgkmcnt(0) -> six MFMAs chained on one
    // accumulator).  Unit u = (k-step s = u / MT_W, M-tile slot mi = u % MT_W); units run in PAIRS so that consecutive MFMAs
    // alternate between two accumulators, and the operand reads of pair P + 1 (and the d_y fragments of a k-step the next
    // pair enters) are issued before the MFMAs of pair P.
    constexpr int NU = 4 * MT_W, G = NT == 1 ? 2 : 1, NP = NU / G;    // NT = 2: the two N tiles already alternate accumulators
    static_assert(NU % G == 0, "units come in groups");
    auto load_a = [&](int u, bf16x8 (&af)[3]) {
      const int s = u / MT_W, mi = u - s * MT_W;
      const int sx = ((s >> 1) * HY * HX + (s & 1) * 4 * HX) * ROWB;      // k-step s: z = s >> 1, rows 4 (s & 1) ..
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = tr_pair(lds, aoff[mi] + p * XPL + sx, aoff[mi] + p * XPL + sx + 2 * HX * ROWB);
    };
    auto load_b = [&](int s, bf16x8 (&bf)[3][NT]) {
      const int sd = s * 32 * ROWB;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          bf[p][n] = tr_pair(lds, boff + (p * NT + n) * DPL + sd, boff + (p * NT + n) * DPL + sd + 2 * TX * ROWB);
    };
    bf16x8 ac[G][3], an[G][3], bb[2][3][NT];
    load_b(0, bb[0]);
#pragma unroll
    for (int g = 0; g < G; ++g) load_a(g, ac[g]);
#pragma unroll
    for (int P = 0; P < NP; ++P) {
      const int ul = G * P + G - 1, sl = ul / MT_W;                      // last unit of this group and its k-step
      if (P + 1 < NP) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int sn = (G * (P + 1) + g) / MT_W;
          if (sn > sl && (g == 0 || (G * (P + 1) + g - 1) / MT_W == sl)) load_b(sn, bb[sn & 1]);   // a k-step the next group enters
        }
#pragma unroll
        for (int g = 0; g < G; ++g) load_a(G * (P + 1) + g, an[g]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g)
        if ((G * P + g) % MT_W == BIAS_MI) {           // the bias tile lives in exactly one (wave, mi) slot
          const u32x4 o4 = {one2, one2, one2, one2}, z4 = {0u, 0u, 0u, 0u};
          ac[g][0] = bias_wave ? __builtin_bit_cast(bf16x8, o4) : ac[g][0];
          ac[g][1] = bias_wave ? __builtin_bit_cast(bf16x8, z4) : ac[g][1];
          ac[g][2] = bias_wave ? __builtin_bit_cast(bf16x8, z4) : ac[g][2];
        }
#define MMG(PA, PB)                                                                                                        \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                                       \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                                    \
          const int u = G * P + g, su = u / MT_W, mu = u - su * MT_W;                                                      \
          acc[mu][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[g][PA], bb[su & 1][PB][n], acc[mu][n], 0, 0, 0);         \
        }
      MMG(2, 0) MMG(0, 2) MMG(1, 1) MMG(1, 0) MMG(0, 1) MMG(0, 0)
#undef MMG
      __builtin_amdgcn_sched_barrier(0);
      if (P + 1 < NP) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int p = 0; p < 3; ++p) ac[g][p] = an[g][p];
      }
    }
  }
  // fragment-major partial tiles: part[bx][by][mt <= MT][n][lane * 4 + reg]
  float* out = a.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)((MT + 1) * NT * 256);
#pragma unroll
  for (int mi = 0; mi < MT_W; ++mi) {
    const int mt = wave + 4 * mi;
    if (mt > MT) continue;
#pragma unroll
    for (int n = 0; n < NT; ++n)
      *reinterpret_cast<float4*>(out + (size_t)(mt * NT + n) * 256 + lane * 4) =
          make_float4(acc[mi][n][0], acc[mi][n][1], acc[mi][n][2], acc[mi][n][3]);
  }
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
  const int slots = p.nt == 1 ? 768 : 512;             // resident workgroups (LDS: 50 / 62 KB)
  int gx = slots / p.gy;
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  p.gx = gx;
  p.mt = (27 * p.nq + 3) / 4;
  p.red_fl = (p.mt + 1) * p.nt * 256;
  return p;
}

}  // namespace

// ---- internal interface for conv3d.hip (C++ linkage, not part of the ABI)
int modetx_wgrad_partials_reduce2(modet_step_ctx* defer, const float* part, float* red, float* dw, float* db, int gx, int gy,
                                  int Cin, int Cout, int nq, int mt, int nt, int n_coblk, hipStream_t s);      // conv3d_bf16.hip
bool modetx_wtr_eligible(int B, int D, int H, int W, int Cin, int Cout) {
  const int64_t n = (int64_t)B * D * H * W;
  return Cin >= 4 && n * (Cin > Cout ? Cin : Cout) * 4 < 0x7fffffffLL && Cout <= 256 && Cin <= 1024;
}
size_t modetx_wtr_ws_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const WtrPlan p = wtr_plan(B, D, H, W, Cin, Cout);
  const int gx = p.gy >= 768 ? 1 : 768 / p.gy;         // upper bound of the plan's gx
  return ((size_t)gx + 1) * p.gy * p.red_fl * sizeof(float);      // workgroup partials + their column sums
}
int modetx_wtr_wgrad(modet_step_ctx* defer, const float* x, const float* dy, float* dw, float* db, void* ws, int B, int D, int H,
                     int W, int Cin, int Cout, hipStream_t s) {
  const WtrPlan p = wtr_plan(B, D, H, W, Cin, Cout);
  WtrArgs a{x, dy, (float*)ws, B, D, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, p.ntiles, p.n_coblk};
  const dim3 grid(p.gx, p.gy);
#define WTR_L(NQ_, NT_) hipLaunchKernelGGL((conv_wgrad_tr_kernel<NQ_, NT_>), grid, dim3(NTHR), 0, s, a)
#define WTR_Q(NT_) do { if (p.nq == 4) WTR_L(4, NT_); else if (p.nq == 3) WTR_L(3, NT_); else if (p.nq == 2) WTR_L(2, NT_); else WTR_L(1, NT_); } while (0)
  if (p.nt == 1) WTR_Q(1); else WTR_Q(2);
#undef WTR_Q
#undef WTR_L
  float* red = (float*)ws + (size_t)p.gx * p.gy * p.red_fl;
  return modetx_wgrad_partials_reduce2(defer, (const float*)ws, red, dw, db, p.gx, p.gy, Cin, Cout, p.nq, p.mt, p.nt, p.n_coblk, s);
}
