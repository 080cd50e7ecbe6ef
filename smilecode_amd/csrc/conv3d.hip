// 3x3x3 / stride 1 / zero-pad 1 convolution as an fp32 MFMA implicit GEMM on gfx950 (exact f32:
// v_mfma_f32_16x16x4_f32 is bitwise an fmaf chain), channels-last activations.
//   reference call sites: nn.Conv3d in ConvBlock / ConvInsBlock / CWM, ModeT/models.py:127,:144,:254
//   (arithmetic itself lives in ATen/MIOpen there).
//
// forward / dgrad:  D[voxel(16 along W)][cout(16)] += A[voxel][cin(4)] * B[cin(4)][cout]   per tap
//   A from an LDS input tile [cin][halo'd voxels] (channel stride = 16 mod 32 dwords -> conflict-free ds_read_b32
//   across the k-groups), B from an LDS weight tile [tap][cin][cout]; a wave owns R rows x NT cout tiles and
//   re-uses each B fragment R times.  dgrad is the same kernel on flipped + transposed weights.
// wgrad:  D[cin(16)][cout(16)] += A[cin][voxel(4)] * B[voxel(4)][cout]   per tap, K runs over the voxels;
//   Cin<=8 packs two taps into the 16 M rows.  Workgroups walk tiles persistently, partial d_w goes to a
//   workspace and is summed in fixed order in fp64 (deterministic, no atomics).
#include "common.h"
#include "step_ctx.h"
#include <cstdlib>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 256;
#ifdef MODET_TUNING
// phase timing of conv3d_mfma_kernel (tools/exp_conv_phases.py): per (workgroup, wave) cycle sums of
// [barrier1, LDS fill, barrier2, flush, prefetch issue, MFMA loop, epilogue staging, stages]
__device__ long long* g_conv_dbg = nullptr;
#define DBG_T(i) if (dbgp) { const long long now_ = clock64(); dsum[i] += now_ - tprev; tprev = now_; }
#else
#define DBG_T(i)
#endif
constexpr int TX = 16, HX = TX + 2;

__host__ __device__ constexpr int pad16mod32(int n) { return ((n - 16 + 31) / 32) * 32 + 16; }

// ------------------------------------------------------------------------------------------------ weight packing
// wpk[tapP][c][n], zero padded to (CinP, CoutP).  W(tap; c -> co) is
//   mode 0 (forward): w[co][c][tap]           w: (Cout, Cin, 27)
//   mode 1 (dgrad)  : w[c][co][26 - tap]      w: (Co=c, Ci=co, 27): a convolution over d_y's channels c
// Row packing P (= 16/CoP output rows share one 16-wide MFMA N tile when Cout <= CoP = 16/P):
//   tapP = (dz*(P+2) + dyp)*3 + dx with dyp = p + dy in [0, P+2);  column n = p*CoP + co holds
//   W((dz, dyp-p, dx); c -> co) when 0 <= dyp-p <= 2, else 0.   P = 1 is the plain layout.
// Layout of conv_direct_kernel (pack modes 2 / 3 = forward / dgrad, P = 1; CinP, CoutP multiples of 16):
//   wpk[tap][cs][ct][k][n][j] = W(tap; c = cs*16 + 4k + j -> co = ct*16 + n)        (k, j in 0..3, n in 0..15)
// i.e. one 1 KB block per (tap, 16 input channels, 16 output channels) in which lane (k, n) of a wave finds the B operands of
// four consecutive MFMAs as ONE float4 (the K order inside a block is a permutation the A operand follows).
__device__ __forceinline__ void pack_direct(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout, int CinP,
                                            int CoutP, int dgrad) {
  const int CS = CinP / 16, CT = CoutP / 16;
  const int total = 27 * CinP * CoutP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 3, n = (i >> 2) & 15, k = (i >> 6) & 3;
    int t = i >> 8;
    const int ct = t % CT; t /= CT;
    const int cs = t % CS, tap = t / CS;
    const int c = cs * 16 + 4 * k + j, co = ct * 16 + n;
    float v = 0.f;
    if (c < Cin && co < Cout)
      v = dgrad ? w[((int64_t)c * Cout + co) * 27 + 26 - tap] : w[((int64_t)co * Cin + c) * 27 + tap];
    wpk[i] = v;
  }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout, int CinP,
                                    int CoutP, int mode, int P) {
  const int ntap = 9 * (P + 2);
  const int total = ntap * CinP * CoutP;
  const int CoP = P > 1 ? 16 / P : CoutP;
  if (mode >= 2) { pack_direct(w, wpk, Cin, Cout, CinP, CoutP, mode - 2); return; }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i % CoutP, t = i / CoutP;
    const int c = t % CinP, tapP = t / CinP;
    const int dx = tapP % 3, dyp = (tapP / 3) % (P + 2), dz = tapP / (3 * (P + 2));
    const int p = P > 1 ? n / CoP : 0, co = P > 1 ? n % CoP : n;
    const int dy = dyp - p;
    float v = 0.f;
    if (c < Cin && co < Cout && dy >= 0 && dy <= 2) {
      const int tap = (dz * 3 + dy) * 3 + dx;
      v = mode == 0 ? w[((int64_t)co * Cin + c) * 27 + tap] : w[((int64_t)c * Cout + co) * 27 + 26 - tap];
    }
    wpk[i] = v;
  }
}

// The same packing for many weight tensors in one launch (modet_conv3d_prepack_*): the jobs travel by value in the
// kernel arguments, blockIdx.y = job.
constexpr int PACK_MAX_JOBS = 48;
struct PackJob { const float* w; float* wpk; int Cin, Cout, CinP, CoutP, mode, P, pad_; };
struct PackTable { PackJob job[PACK_MAX_JOBS]; int n; };
__global__ void pack_weights_many_kernel(const PackTable t) {
  const PackJob& J = t.job[blockIdx.y];
  const float* __restrict__ w = J.w;
  float* __restrict__ wpk = J.wpk;
  const int Cin = J.Cin, Cout = J.Cout, CinP = J.CinP, CoutP = J.CoutP, mode = J.mode, P = J.P;
  const int ntap = 9 * (P + 2);
  const int total = ntap * CinP * CoutP;
  const int CoP = P > 1 ? 16 / P : CoutP;
  if (mode >= 2) { pack_direct(w, wpk, Cin, Cout, CinP, CoutP, mode - 2); return; }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i % CoutP, tt = i / CoutP;
    const int c = tt % CinP, tapP = tt / CinP;
    const int dx = tapP % 3, dyp = (tapP / 3) % (P + 2), dz = tapP / (3 * (P + 2));
    const int p = P > 1 ? n / CoP : 0, co = P > 1 ? n % CoP : n;
    const int dy = dyp - p;
    float v = 0.f;
    if (c < Cin && co < Cout && dy >= 0 && dy <= 2) {
      const int tap = (dz * 3 + dy) * 3 + dx;
      v = mode == 0 ? w[((int64_t)co * Cin + c) * 27 + tap] : w[((int64_t)c * Cout + co) * 27 + 26 - tap];
    }
    wpk[i] = v;
  }
}

// Optional input transform of the forward kernel ("lazy InstanceNorm", inference): x is the RAW output of a
// ConvInsBlock whose per-(sample, channel) mean / rstd are given; LeakyReLU((x - mean) * rstd) is applied to the
// prefetched registers right before they are written to LDS (zero padding stays zero), so the normalised tensor never
// exists in HBM.  A thread's prefetch slots all belong to one channel group (slot index = tid + i*NTHR,
// NTHR % (CK/4) == 0), so the transform needs one float4 of mean and rstd per thread and stage.
// (The same transform in the weight-gradient kernels was measured too: +60 us per level-1/2 layer, more than the
// apply pass it saves, so training keeps the materialised tensor.)
struct ConvIn {
  const float* mean;
  const float* rstd;
  int stats_rows;      // rows per sample of the statistics buffer (>= gridDim.x; the tail is zero-filled by workgroup 0)
  const float* shift;  // [sample][Cout] shift K of the fused statistics (sums of (y - K), (y - K)^2); see conv_shift_kernel
};

// Shift of the fused InstanceNorm statistics.  The epilogue accumulates sum(y - K) and sum((y - K)^2) in fp32 and the
// finalize (norm_act.hip) rebuilds mean = K + s1/V, var = s2/V - (s1/V)^2 in fp64.  K[b][co] is the conv output at voxel
// (1,1,1) of sample b -- for skull-stripped volumes (exact zeros outside the brain, reference makePklDataset.py:19-20) that
// is the constant every background voxel produces, so more than half of the summands become exact zeros.  Without the
// shift those identical summands round the SAME way every time they are added to a growing fp32 accumulator: a coherent
// (not random) error of ~1e-5 in sum(y^2), which var = E[y^2] - E[y]^2 then amplifies by E[y^2]/var (~100 for the first
// ConvInsBlock, whose input is a LeakyReLU of a non-negative image): 1.5e-4 on the normalised tensor at 160x192x160
// against 1e-5 for ATen (profiles/r02_attrib_fullsize_before.json), and 5-9e-3 voxels on the final flow.
// K only has to be NEAR the data (any value is exact in exact arithmetic), so this kernel may sum in any order:
// one wave per (sample, cout), lanes over (tap, cin).
__global__ __launch_bounds__(256) void conv_shift_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ in_mean,
                                                         const float* __restrict__ in_rstd, float* __restrict__ shift, int B,
                                                         int D, int H, int W, int Cin, int Cout) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);          // (b, co)
  if (item >= B * Cout) return;
  const int b = item / Cout, co = item - b * Cout;
  const int z = D > 1 ? 1 : 0, y = H > 1 ? 1 : 0, xx = W > 1 ? 1 : 0;
  float acc = 0.f;
  const int n = 27 * Cin;
  const float* xb = x + (int64_t)b * D * H * W * Cin;
  // four (tap, cin) items per lane and trip, their loads issued together (clamped addresses, fenced): this launch is a chain of
  // dependent round trips in front of every statistics conv -- one load per `continue` test made it ~10 us for ~nothing
  for (int i0 = lane; i0 < n; i0 += 256) {
    float xv[4], wv[4], mv[4], rv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 64 * u;
      const int ii = i < n ? i : 0;
      const int tap = ii / Cin, c = ii - tap * Cin;
      const int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, xq = xx + tap % 3 - 1;
      ok[u] = i < n && zz >= 0 && zz < D && yy >= 0 && yy < H && xq >= 0 && xq < W;
      xv[u] = xb[ok[u] ? (((int64_t)zz * H + yy) * W + xq) * Cin + c : 0];
      wv[u] = w[((int64_t)co * Cin + c) * 27 + tap];
      mv[u] = in_mean ? in_mean[b * Cin + c] : 0.f;
      rv[u] = in_mean ? in_rstd[b * Cin + c] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(xv[u]), "+v"(wv[u]), "+v"(mv[u]), "+v"(rv[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = xv[u];
      if (in_mean) v = lrelu((v - mv[u]) * rv[u]);
      if (ok[u]) acc = fmaf(v, wv[u], acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) shift[item] = acc + (bias ? bias[co] : 0.f);
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
// Persistent workgroups walk (tile, Cin-chunk) stages.  Software pipeline: the global loads of stage s+1 (input
// tile with halo and, when Cin spans several chunks, the weight slab) are issued into registers right after
// stage s has been written to LDS, i.e. before the MFMA loop of stage s, so HBM/L2 latency hides under the
// matrix work (PMC before this change: MFMA pipe 54 % busy, 36 % of wave cycles in s_waitcnt/barrier).
// XF ("extra features", inference): the lazy-InstanceNorm input transform (ConvIn) and statistics from the direct-store
// epilogue.  Compiled out of the kernels training uses: carrying them as run-time options cost those 4 %.
// ST: the staged epilogue also produces the fused InstanceNorm statistics (forward convs of ConvInsBlocks with
// Cout 4/8/16).  A compile-time switch: as a run-time option the statistics registers and the shift load cost the
// kernels that never use them (dgrad, plain convs) 4-5 % (tools/ab_kernels.py, r01 vs r02a builds on one box).
template <int TZ, int TY, int WM, int WN, int NT, int CK, bool VEC4, int P, bool MULTI, bool XF, bool ST = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 && CK == 8 && !XF && P >= 2) ? 4 : 1) void conv3d_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                           const float* __restrict__ bias, float* __restrict__ y, int D,
                                                           int H, int W, int Cin, int Cout, int CinP, int CoutP, int act,
                                                           int tiles_x, int tiles_y, int tiles_z, int ntiles, float* __restrict__ stats, ConvIn inorm) {
  constexpr int NTHR = WM * WN * 64;         // 4 or 8 waves per workgroup (shadows the file-level constant)
  static_assert(P == 1 || (WN == 1 && NT == 1 && TY % P == 0), "row packing needs a single 16-wide N tile");
  constexpr int ROWS = TZ * TY, NCB = WN * NT * 16;
  constexpr int R = ROWS / P / WM;           // row groups (P output rows each) per wave
  constexpr int NTAP = 9 * (P + 2), CoP = 16 / P;
  constexpr int HZ = TZ + 2, HY = TY + 2, HVOX = HZ * HY * HX;
  constexpr int CS = pad16mod32(HVOX);
  constexpr int NCBS = (NCB % 32 == 16) ? NCB : NCB + 16;
  constexpr int QX = CK / 4, NXV = (HVOX * QX + NTHR - 1) / NTHR;
  constexpr int QW = NCB / 4, NWV = (NTAP * CK * QW + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) float xs[CK * CS];
  __shared__ __attribute__((aligned(16))) float wsm[NTAP * CK * NCBS];
  // output staging [voxel][OC], one private slice per wave (rows it computed), flushed one iteration later
  constexpr int OC = P > 1 ? CoP : NCB;
  constexpr bool STG = (P > 1) || (NCB == 16);                     // the L1/L2 configs (16-wide N tile)
  __shared__ __attribute__((aligned(16))) float stg[STG ? ROWS * TX * OC : 4];
  constexpr bool HAS_STATS = XF || ST;       // statistics code exists in this instantiation
  // statistics shift K of the CURRENT sample for this block's <= 16 channels (staged epilogue): read back with the
  // staging tile's own LDS reads, so the flush waits for no extra global load (see ConvIn)
  __shared__ __attribute__((aligned(16))) float ksm[(HAS_STATS && STG) ? 16 : 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lk = lane >> 4;
  const int cb0 = blockIdx.y * NCB;
  constexpr bool vec4 = VEC4;
  constexpr bool multi = MULTI;              // weights change per stage only when Cin spans several chunks (CinP > CK)

  int tile = blockIdx.x;
  if (tile >= ntiles) return;

  float4 xr[NXV], wr[MULTI ? NWV : 1];

  // tile-invariant part of the fill, computed once per thread: halo coordinates (packed), the LDS address and the
  // global offset of each prefetch slot relative to the tile origin
  int xh[NXV], xrel[NXV];
#pragma unroll
  for (int i = 0; i < NXV; ++i) {
    const int idx = tid + i * NTHR;
    const bool on = idx < HVOX * QX;
    const int hv = on ? idx / QX : 0, c4 = on ? idx - hv * QX : 0;
    const int hx = hv % HX, t2 = hv / HX;
    const int hy = t2 % HY, hz = t2 / HY;
    xh[i] = on ? (hz | (hy << 8) | (hx << 16) | (c4 << 24)) : -1;
    xrel[i] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + c4 * 4;       // |.| < 3*H*W*Cin: fits 32 bits
  }

  float4 im4 = make_float4(0.f, 0.f, 0.f, 0.f), ir4 = make_float4(1.f, 1.f, 1.f, 1.f);   // lazy InstanceNorm of the input
  unsigned vmask = 0;                                    // which prefetch slots hold in-volume data (NXV <= 32)
  static_assert(NXV <= 32, "slot mask");
  auto load_stage = [&](int tl, int c0, bool with_w) {
    int t = tl;
    const int x0 = (t % tiles_x) * TX; t /= tiles_x;
    const int y0 = (t % tiles_y) * TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * TZ;
    const float* xt = x + (((int64_t)(t / tiles_z) * D + z0) * H + y0) * W * Cin + (int64_t)x0 * Cin + c0;
    if (XF && inorm.mean) {
      const int cg = c0 + (tid % QX) * 4;
      if (cg < Cin) {
        im4 = *reinterpret_cast<const float4*>(inorm.mean + (t / tiles_z) * Cin + cg);
        ir4 = *reinterpret_cast<const float4*>(inorm.rstd + (t / tiles_z) * Cin + cg);
      }
      vmask = 0;
    }
    // a tile whose halo lies inside the volume (70 % of them at 160x192x160) and a full channel chunk: plain loads
    const bool interior = vec4 && z0 > 0 && z0 + TZ < D && y0 > 0 && y0 + TY < H && x0 > 0 && x0 + TX < W && c0 + CK <= Cin;
    if (interior) {
#pragma unroll
      for (int i = 0; i < NXV; ++i)
        xr[i] = xh[i] >= 0 ? *reinterpret_cast<const float4*>(xt + xrel[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (XF) vmask = 0xffffffffu;
    } else
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xh[i] >= 0) {
        const int z = z0 + (xh[i] & 255) - 1, yy = y0 + ((xh[i] >> 8) & 255) - 1, xx = x0 + ((xh[i] >> 16) & 255) - 1;
        const int c = c0 + (xh[i] >> 24) * 4;
        if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W && c < Cin) {
          const float* p = xt + xrel[i];
          if (XF) vmask |= 1u << i;
          if (vec4) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (c + 1 < Cin) v.y = p[1];
            if (c + 2 < Cin) v.z = p[2];
            if (c + 3 < Cin) v.w = p[3];
          }
        }
      }
      xr[i] = v;
    }
    if constexpr (MULTI) {
      if (with_w) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
          const int idx = tid + i * NTHR;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < NTAP * CK * QW) {
            const int n4 = idx % QW, row = idx / QW;
            const int tap = row / CK, cc = row - tap * CK;
            v = *reinterpret_cast<const float4*>(wpk + ((int64_t)(tap * CinP + c0 + cc)) * CoutP + cb0 + n4 * 4);
          }
          wr[i] = v;
        }
      }
    }
  };

  auto store_stage = [&](bool with_w) {
    if (XF && inorm.mean) {
#pragma unroll
      for (int i = 0; i < NXV; ++i) {
        if ((vmask >> i) & 1u) {
          xr[i].x = lrelu((xr[i].x - im4.x) * ir4.x); xr[i].y = lrelu((xr[i].y - im4.y) * ir4.y);
          xr[i].z = lrelu((xr[i].z - im4.z) * ir4.z); xr[i].w = lrelu((xr[i].w - im4.w) * ir4.w);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      if (xh[i] >= 0) {
        const int l = ((xh[i] >> 24) * 4) * CS + ((xh[i] & 255) * HY + ((xh[i] >> 8) & 255)) * HX + ((xh[i] >> 16) & 255);
        xs[l] = xr[i].x;
        xs[l + CS] = xr[i].y;
        xs[l + 2 * CS] = xr[i].z;
        xs[l + 3 * CS] = xr[i].w;
      }
    }
    if constexpr (MULTI) {
      if (with_w) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
          const int idx = tid + i * NTHR;
          if (idx < NTAP * CK * QW) {
            const int n4 = idx % QW, row = idx / QW;
            *reinterpret_cast<float4*>(wsm + row * NCBS + n4 * 4) = wr[i];
          }
        }
      }
    }
  };

  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int rowbase[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int rr = (wm * R + r) * P;         // first output row of the group
    rowbase[r] = ((rr / TY) * HY + (rr % TY)) * HX + li + lk * CS;
  }

  // output staging through LDS when the [voxel][NCB] tile fits in the input-tile buffer and Cout is float4-able
  // staged path needs float4-able channel vectors whose group index is lane-invariant (64 % (Cout/4) == 0)
  const bool lds_epi_rt = STG && ((Cout & 3) == 0) && (Cout <= OC) && (64 % (Cout >> 2) == 0);
  // bias for the LDS-staged epilogue: a lane always stores channel group c4 = lane % (Cout/4), so its bias float4 is
  // loaded ONCE here.  A load inside the store loop would put an s_waitcnt vmcnt(0) in
  // front of every store, and vmcnt also counts stores: the epilogue would wait for each store's acknowledgement.
  float4 bq4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lds_epi_rt && bias) bq4 = *reinterpret_cast<const float4*>(bias + cb0 + (lane % (Cout >> 2)) * 4);
  float bv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = P > 1 ? li % CoP : cb0 + (wn * NT + n) * 16 + li;
    bv[n] = (bias && co < Cout) ? bias[co] : 0.f;
  }
  const int tiles_per_b = tiles_x * tiles_y * tiles_z;
  if (HAS_STATS && STG && stats && lds_epi_rt && tid < 16)        // shift of the first tile's sample (visible after the
    ksm[tid] = cb0 + tid < Cout ? inorm.shift[(tile / tiles_per_b) * Cout + cb0 + tid] : 0.f;   // loop's first barrier)

  if constexpr (!MULTI) {
    for (int idx = tid; idx < NTAP * CK * QW; idx += NTHR) {
      const int n4 = idx % QW, row = idx / QW;
      const int tap = row / CK, cc = row - tap * CK;
      *reinterpret_cast<float4*>(wsm + row * NCBS + n4 * 4) =
          *reinterpret_cast<const float4*>(wpk + ((int64_t)(tap * CinP + cc)) * CoutP + cb0 + n4 * 4);
    }
  }
  // deferred flush of a finished tile: the wave reads back ITS OWN staging slice (rows it computed) and writes whole
  // channel vectors as float4.  Issued at the start of the next iteration, BEFORE that iteration's prefetch loads:
  // s_waitcnt vmcnt is in-order and counts stores, so with stores issued right before the wait (as a same-iteration
  // epilogue does) every tile would wait for a full store round trip.
  int ptile = -1;
  // per-lane constants of the flush: item i = lane + 64*it of this wave's WROWS x TX x (Cout/4) float4s decomposes into
  // (row, voxel, channel group) once per kernel (runtime divisions by Cout/4), not once per tile
  constexpr int WROWS = R * P;                         // rows owned by this wave
  constexpr int FL_MAX = (WROWS * TX * (OC / 4) + 63) / 64;
  int fl_stg[FL_MAX], fl_pos[FL_MAX], fl_rel[FL_MAX];
  {
    const int cq = lds_epi_rt ? (Cout >> 2) : 1, per_row = TX * cq;
#pragma unroll
    for (int it = 0; it < FL_MAX; ++it) {
      const int i = lane + 64 * it;
      const bool on = lds_epi_rt && i < WROWS * per_row;
      const int rl = on ? i / per_row : 0, f = on ? i - rl * per_row : 0;
      const int rr = wm * WROWS + rl;
      const int vx = f / cq, c4 = f - vx * cq;
      fl_stg[it] = on ? (rr * TX + vx) * OC + c4 * 4 : -1;
      fl_pos[it] = (rr / TY) | ((rr % TY) << 8) | (vx << 16);
      fl_rel[it] = (((rr / TY) * H + (rr % TY)) * W + vx) * Cout + c4 * 4;
    }
  }
  // fused InstanceNorm statistics: per-lane sums live in registers across ALL tiles of one batch sample this
  // workgroup walks (tiles are sample-major, so the sample index only ever increases); the cross-lane / cross-wave
  // reduction and the store happen once per (workgroup, sample): stats[sample][workgroup][channel][2]
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  int stat_b = -1;
  unsigned stat_done = 0;                              // samples this workgroup already wrote (B <= 32)
  // called by every wave at the same point (the sample index changes for the whole workgroup at once): wave sums ->
  // LDS -> one row per (sample, workgroup) in fixed wave order
  float dsx[NT], dsq[NT];                              // direct-store epilogue: this lane's cout per n tile
#pragma unroll
  for (int n = 0; n < NT; ++n) { dsx[n] = 0.f; dsq[n] = 0.f; }
  // The per-wave partials go through LDS that is free at the time of the call -- the output staging for the staged
  // epilogue (emission happens after a flush), the input tile for the direct-store epilogue (after the MFMA loop) -- so
  // the statistics add no LDS (an extra 1 KB array pushed two configurations over an occupancy step).
  constexpr int SRW = NT * 16 * 2 > 32 ? NT * 16 * 2 : 32;
  auto emit_stats = [&](int bsamp) {
    float* sredp = lds_epi_rt ? stg : xs;
    __syncthreads();                                   // every wave is done with that buffer
    if (XF && !lds_epi_rt) {
      // direct-store configs: lane = (cout li, voxel group lk); sum over lk, then over the WM waves of this wn
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        dsx[n] += __shfl_xor(dsx[n], 16, 64); dsq[n] += __shfl_xor(dsq[n], 16, 64);
        dsx[n] += __shfl_xor(dsx[n], 32, 64); dsq[n] += __shfl_xor(dsq[n], 32, 64);
        if (lane < 16) { sredp[wave * SRW + (n * 16 + lane) * 2] = dsx[n]; sredp[wave * SRW + (n * 16 + lane) * 2 + 1] = dsq[n]; }
        dsx[n] = 0.f; dsq[n] = 0.f;
      }
      __syncthreads();
      if (tid < NCB * 2) {
        const int cc = tid >> 1, wnc = cc / (NT * 16), within = (cc % (NT * 16)) * 2 + (tid & 1);
        float acc_s = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < WM; ++w8) acc_s += sredp[(w8 * WN + wnc) * SRW + within];
        if (cb0 + cc < Cout) stats[(((int64_t)bsamp * inorm.stats_rows + blockIdx.x) * Cout + cb0 + cc) * 2 + (tid & 1)] = acc_s;
      }
      __syncthreads();
      stat_done |= 1u << bsamp;
      return;
    }
    const int cq = Cout >> 2;
    for (int o = cq; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sx[j] += __shfl_xor(sx[j], o, 64); sq[j] += __shfl_xor(sq[j], o, 64); }
    }
    if (lane < cq) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sredp[wave * SRW + (lane * 4 + j) * 2] = sx[j]; sredp[wave * SRW + (lane * 4 + j) * 2 + 1] = sq[j]; }
    }
    __syncthreads();
    if (tid < 2 * Cout) {
      float acc_s = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < WM; ++w8) acc_s += sredp[w8 * WN * SRW + tid];  // staged configs have WN == 1
      stats[((int64_t)bsamp * inorm.stats_rows + blockIdx.x) * Cout * 2 + tid] = acc_s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { sx[j] = 0.f; sq[j] = 0.f; }
    stat_done |= 1u << bsamp;
  };
  auto flush_tile = [&](int next_tile) {                // next_tile: the tile the NEXT flush will write, -1 = none
    if (ptile < 0) return;
    int t = ptile;
    const int x0 = (t % tiles_x) * TX; t /= tiles_x;
    const int y0 = (t % tiles_y) * TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * TZ;
    const int64_t xbase = (int64_t)(t / tiles_z) * D * H * W;
    const int cq = Cout >> 2;
    float* ytile = y + (xbase + ((int64_t)z0 * H + y0) * W + x0) * Cout + cb0;
    // statistics shift of this lane's channel group (see ConvIn): a transient of the flush, NOT held across the MFMA loop
    // (four more persistent registers cost several configurations a wave of occupancy), read from LDS
    float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_STATS && stats) {
      stat_b = t / tiles_z;
      k4 = *reinterpret_cast<const float4*>(ksm + (lane % cq) * 4);
    }
#pragma unroll
    for (int it = 0; it < FL_MAX; ++it) {
      if (fl_stg[it] < 0) continue;
      const int z = z0 + (fl_pos[it] & 255), yy = y0 + ((fl_pos[it] >> 8) & 255), xx = x0 + (fl_pos[it] >> 16);
      if (z < D && yy < H && xx < W) {
        float4 v = *reinterpret_cast<const float4*>(stg + fl_stg[it]);
        v.x += bq4.x; v.y += bq4.y; v.z += bq4.z; v.w += bq4.w;
        if (HAS_STATS && stats) {       // InstanceNorm statistics of the conv output, fused (act == 0 on this path)
          const float e0 = v.x - k4.x, e1 = v.y - k4.y, e2 = v.z - k4.z, e3 = v.w - k4.w;
          sx[0] += e0; sx[1] += e1; sx[2] += e2; sx[3] += e3;
          sq[0] = fmaf(e0, e0, sq[0]); sq[1] = fmaf(e1, e1, sq[1]);
          sq[2] = fmaf(e2, e2, sq[2]); sq[3] = fmaf(e3, e3, sq[3]);
        }
        if (act) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
        *reinterpret_cast<float4*>(ytile + fl_rel[it]) = v;
      }
    }
    ptile = -1;
    // last tile of this sample for this workgroup?  emit now, while the staging buffer is free
    if (HAS_STATS && stats && next_tile >= 0 && next_tile / tiles_per_b != stat_b) {
      emit_stats(stat_b);              // ends with a barrier: every wave has finished reading ksm for this sample
      stat_b = -1;
      if (tid < 16) ksm[tid] = cb0 + tid < Cout ? inorm.shift[(next_tile / tiles_per_b) * Cout + cb0 + tid] : 0.f;
    }
  };

  int c0 = 0;
  bool first = true;
#ifdef MODET_TUNING
  long long* dbgp = g_conv_dbg;
  long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#endif
  load_stage(tile, 0, true);
  while (true) {
    DBG_T(7)
    __syncthreads();                               // every wave is done reading the previous stage from LDS
    DBG_T(0)
    store_stage(multi || first);
    DBG_T(1)
    __syncthreads();
    DBG_T(2)
    int ntile = tile, nc0 = c0 + CK;
    bool has_next = true;
    if (nc0 >= CinP) { nc0 = 0; ntile = tile + gridDim.x; has_next = ntile < ntiles; }
    if (STG) flush_tile(tile);                     // previous tile's stores go out before this prefetch
    DBG_T(3)
    if (has_next) load_stage(ntile, nc0, multi);   // in flight during the MFMA loop below
    DBG_T(4)
    first = false;

    __builtin_amdgcn_s_setprio(1);
    {
      // K loop over (tap, 4-channel group), fully unrolled: every LDS address is base + immediate.  The A/B
      // fragments of step ks+1 are read into a second register set before the MFMAs of step ks issue.
      constexpr int KG = CK / 4, KS = NTAP * KG;
      float af[2][R], bfr[2][NT];
      auto frag_load = [&](int ks, float (&a)[R], float (&bq)[NT]) {
        const int tap = ks / KG, kg = ks - tap * KG;
        const int dz = tap / (3 * (P + 2)), dy = (tap / 3) % (P + 2), dx = tap % 3;   // dy = p + dy_orig when packed
        const int toff = (dz * HY + dy) * HX + dx + kg * 4 * CS;
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[n] = wsm[(tap * CK + kg * 4 + lk) * NCBS + (wn * NT + n) * 16 + li];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = xs[rowbase[r] + toff];
      };
      frag_load(0, af[0], bfr[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) frag_load(ks + 1, af[(ks + 1) & 1], bfr[(ks + 1) & 1]);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][r], bfr[ks & 1][n], acc[r][n], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    DBG_T(5)

    if (c0 + CK >= CinP) {
      int t = tile;
      const int x0 = (t % tiles_x) * TX; t /= tiles_x;
      const int y0 = (t % tiles_y) * TY; t /= tiles_y;
      const int z0 = (t % tiles_z) * TZ;
      const int64_t xbase = (int64_t)(t / tiles_z) * D * H * W;
      if (STG && lds_epi_rt) {
        // write this wave's accumulators into its private staging slice; the global stores happen next iteration
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int rr = (wm * R + r) * P + (P > 1 ? li / CoP : 0);
            const int cs = P > 1 ? li % CoP : (wn * NT + n) * 16 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j) stg[(rr * TX + lk * 4 + j) * OC + cs] = acc[r][n][j];
            acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
        ptile = tile;
      } else {
        // direct stores: lane holds (row p, cout) = li, voxels x = lk*4 + j; bias was hoisted out of the stage loop
        if (XF && stats) {
          const int bs = t / tiles_z;
          if (bs != stat_b) {
            if (stat_b >= 0) emit_stats(stat_b);
            stat_b = bs;
          }
        }
        float kv[NT];                                          // statistics shift of this lane's couts (see ConvIn)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = P > 1 ? li % CoP : cb0 + (wn * NT + n) * 16 + li;
          kv[n] = (XF && stats && co < Cout) ? inorm.shift[(t / tiles_z) * Cout + co] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int rr = (wm * R + r) * P + (P > 1 ? li / CoP : 0);
          const int z = z0 + rr / TY, yy = y0 + rr % TY;
          float* yrow = y + (xbase + ((int64_t)z * H + yy) * W + x0 + lk * 4) * Cout;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int co = P > 1 ? li % CoP : cb0 + (wn * NT + n) * 16 + li;
            if (z < D && yy < H && co < Cout) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (x0 + lk * 4 + j < W) {
                  float v = acc[r][n][j] + bv[n];
                  if (XF && stats) { const float e = v - kv[n]; dsx[n] += e; dsq[n] = fmaf(e, e, dsq[n]); }
                  if (act) v = lrelu(v);
                  yrow[j * Cout + co] = v;
                }
              }
            }
            acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
    DBG_T(6)
    if (!has_next) break;
    tile = ntile;
    c0 = nc0;
  }
  if (STG) {
    __syncthreads();
    flush_tile(-1);
  }
  if (HAS_STATS && stats && (XF || lds_epi_rt)) {
    if (stat_b >= 0) emit_stats(stat_b);
    const int nb = ntiles / (tiles_x * tiles_y * tiles_z);
    for (int bsamp = 0; bsamp < nb; ++bsamp)             // samples this workgroup never touched: zero rows
      if (!((stat_done >> bsamp) & 1u)) emit_stats(bsamp);
    if (blockIdx.x == 0) {                               // rows the buffer has beyond this launch's grid
      const int tail = (inorm.stats_rows - (int)gridDim.x) * Cout * 2;
      for (int bsamp = 0; bsamp < nb; ++bsamp)
        for (int i = tid; i < tail; i += NTHR)
          if (i % (Cout * 2) >= cb0 * 2 && i % (Cout * 2) < (cb0 + NCB) * 2)
            stats[((int64_t)bsamp * inorm.stats_rows + gridDim.x) * Cout * 2 + i] = 0.f;
    }
  }
#ifdef MODET_TUNING
  if (dbgp && lane == 0) {
    long long* o = dbgp + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * (NTHR / 64) + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = dsum[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------------ wgrad
constexpr int WG_TY = 8, WG_HY = WG_TY + 2;         // wgrad voxel tile: WG_TZ (template) x 8 x 16

template <int CIT, int WG_TZ>   // CIT: input channels per tap in the 16 M rows (16 -> 1 tap, 8 -> 2, 4 -> 4 taps per MFMA)
__global__ __launch_bounds__(NTHR) void conv3d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ part, int D, int H, int W, int Cin,
                                                            int Cout, int tiles_x, int tiles_y, int tiles_z,
                                                            int ntiles, int n_ci_tiles) {
  // tap index 27 is always a free slot (27 taps + 1 = 28 = 4 waves x 7 = 7 groups x 4 = 14 groups x 2): it carries
  // the bias gradient (A = 1), so d_bias costs no extra pass over d_y
  constexpr int WG_ROWS = WG_TZ * WG_TY, WG_HVOX = (WG_TZ + 2) * WG_HY * HX;
  constexpr int TP = 16 / CIT, NG = (28 + TP - 1) / TP, GPW = (NG + 3) / 4, NSLOT = NG;
  __shared__ __attribute__((aligned(16))) float xs[WG_HVOX * CIT];
  __shared__ __attribute__((aligned(16))) float dys[WG_ROWS * TX * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int ci_tile = blockIdx.y % n_ci_tiles, co_tile = blockIdx.y / n_ci_tiles;
  const int ci0 = ci_tile * CIT, co0 = co_tile * 16;

  // No per-MFMA select: groups past NG (CIT = 8 only) multiply real tile data into accumulators nobody reads, and
  // the bias slot (tap 27: A = 1) lives in exactly one (wave, g) = (BIAS_W, BIAS_G), so only that g carries a
  // v_cndmask (a select in front of every MFMA cost 11-14 % of the kernel: VALU issue + the VALU->MFMA hazard nops).
  constexpr int BIAS_GRP = 27 / TP, BIAS_W = BIAS_GRP / GPW, BIAS_G = BIAS_GRP % GPW;
  f32x4 acc[GPW];
  int aoff[GPW];
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int grp = wave * GPW + g;
    const int tap = grp * TP + li / CIT;
    const int tt = tap < 27 ? tap : 0;
    const int dz = tt / 9, dyy = (tt / 3) % 3, dx = tt % 3;
    aoff[g] = ((dz * WG_HY + dyy) * HX + dx + lk) * CIT + (li % CIT);
  }
  const bool bsel = (wave == BIAS_W) && (li / CIT == 27 % TP);        // rows of tap 27: d_bias = sum dy * 1

  // software pipeline over the tiles this workgroup owns: next tile's x / d_y loads are issued into registers
  // before the MFMA loop of the current tile
  constexpr int QX = CIT / 4, NXV = (WG_HVOX * QX + NTHR - 1) / NTHR;
  constexpr int NDV = (WG_ROWS * TX * 4) / NTHR;
  float4 xr[NXV], dr[NDV];
  const bool xvec = (Cin & 3) == 0, dvec = (Cout & 3) == 0;

  auto load_tile = [&](int tl) {
    int t = tl;
    const int x0 = (t % tiles_x) * TX; t /= tiles_x;
    const int y0 = (t % tiles_y) * WG_TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * WG_TZ;
    const int64_t vb = (int64_t)(t / tiles_z) * D * H * W;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int idx = tid + i * NTHR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < WG_HVOX * QX) {
        const int hv = idx / QX, c4 = idx - hv * QX;
        const int hx = hv % HX, t2 = hv / HX;
        const int hy = t2 % WG_HY, hz = t2 / WG_HY;
        const int z = z0 + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
        const int c = ci0 + c4 * 4;
        if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W && c < Cin) {
          const float* p = x + (vb + ((int64_t)z * H + yy) * W + xx) * Cin + c;
          if (xvec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (c + 1 < Cin) v.y = p[1];
            if (c + 2 < Cin) v.z = p[2];
            if (c + 3 < Cin) v.w = p[3];
          }
        }
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int idx = tid + i * NTHR;
      const int vox = idx >> 2, c4 = idx & 3;
      const int row = vox / TX, xx = x0 + vox % TX;
      const int z = z0 + row / WG_TY, yy = y0 + row % WG_TY;
      const int co = co0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z < D && yy < H && xx < W && co < Cout) {
        const float* p = dy + (vb + ((int64_t)z * H + yy) * W + xx) * Cout + co;
        if (dvec) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (co + 1 < Cout) v.y = p[1];
          if (co + 2 < Cout) v.z = p[2];
          if (co + 3 < Cout) v.w = p[3];
        }
      }
      dr[i] = v;
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < WG_HVOX * QX) *reinterpret_cast<float4*>(xs + idx * 4) = xr[i];     // [voxel][CIT], idx = hv*QX + c4
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) *reinterpret_cast<float4*>(dys + (tid + i * NTHR) * 4) = dr[i];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll 16
    for (int row = 0; row < WG_ROWS; ++row) {
      const int rb = ((row / WG_TY) * WG_HY + (row % WG_TY)) * HX * CIT;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bf = dys[(row * TX + s * 4 + lk) * 16 + li];
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
          float a = xs[rb + s * 4 * CIT + aoff[g]];
          if (g == BIAS_G) a = bsel ? 1.f : a;
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf, acc[g], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  // partial[(bx*gridDim.y + by)*NG + grp][i = M row][col = cout]
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    const int grp = wave * GPW + g;
    if (grp >= NG) continue;
    float* p = part + (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * NG + grp) * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) p[(lk * 4 + j) * 16 + li] = acc[g][j];
  }
}

// d_w / d_bias = sum over the gx workgroups of their partial tiles.  One 256-thread workgroup per quarter tile (64
// consecutive floats): wave p adds the partials bx = p, p+4, ... (each load is a coalesced 256 B row segment, eight
// independent fp64 accumulators), the four waves are combined in fixed order through LDS, then every lane scatters
// its element to d_w[co][ci][tap] / d_bias[co].  Fixed assignment + fixed order: deterministic.
//   MODE 0: tiles of conv3d_wgrad_kernel     part[(bx*gy + by)*ng + grp][mrow][col]
//   MODE 1: tiles of conv3d_wgrad_np_kernel  part[bx*ng + grp][mrow][(q, co)]        (gy == 1)
template <int MODE>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw,
                                                  float* __restrict__ db, int Cin, int Cout, int gx, int gy,
                                                  int n_ci_tiles, int cit, int ng, int blk, double (*sm)[64]) {
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int quarter = blk & 3, tl = blk >> 2;                     // tl = by*ng + grp
  const int by = tl / ng, grp = tl - by * ng;
  const float* pp = part + (int64_t)tl * 256 + quarter * 64 + lane;
  const int64_t st = (int64_t)gy * ng * 256;
  // The long jobs (512-1024 partial rows, 10-28 tiles) are chains of dependent batches: sixteen rows in flight per wave,
  // every batch fenced so that its loads are issued together (the tail too: it was one load -> wait -> add per row).
  double a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.0;
  int bx = ph;
  for (; bx + 60 < gx; bx += 64) {
    float r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = pp[(int64_t)(bx + 4 * u) * st];
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(r[u]));
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] += (double)r[u];
  }
  {
    float r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int b2 = bx + 4 * u;
      r[u] = pp[(int64_t)(b2 < gx ? b2 : 0) * st];               // row 0 always exists
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(r[u]));
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] += bx + 4 * u < gx ? (double)r[u] : 0.0;
  }
  double v = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
             (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
  if (ph > 0) sm[ph - 1][lane] = v;
  __syncthreads();
  if (ph > 0) return;
  v = ((v + sm[0][lane]) + sm[1][lane]) + sm[2][lane];
  const int e = quarter * 64 + lane, mrow = e >> 4, col = e & 15;
  if (MODE == 0) {
    const int tp = 16 / cit;
    const int tap = grp * tp + mrow / cit;
    const int ci = (by % n_ci_tiles) * cit + mrow % cit, co = (by / n_ci_tiles) * 16 + col;
    if (co >= Cout) return;
    if (tap < 27) {
      if (ci < Cin) dw[((int64_t)co * Cin + ci) * 27 + tap] = (float)v;
    } else if (tap == 27 && db && mrow % cit == 0 && by % n_ci_tiles == 0) {
      db[co] = (float)v;
    }
  } else {
    const int cpg = 8 / cit, ngt = ng - 1;
    const int q = col >> 3, co = col & 7;
    if (co >= Cout) return;
    if (grp == ngt) {
      if (db && mrow == 0 && q == 0) db[co] = (float)v;
      return;
    }
    const int tq = mrow / cit, ci = mrow % cit;              // tq = (combo within group)*2 + t
    const int combo = grp * cpg + (cpg == 2 ? (tq >> 1) : 0), t = tq & 1;
    // (t, q): dx = -1 <-> (0,1), 0 <-> (0,0), +1 <-> (1,0); (1,1) is not a tap
    if (combo >= 9 || ci >= Cin || (t == 1 && q == 1)) return;
    const int dxi = t == 1 ? 2 : (q == 1 ? 0 : 1);
    dw[((int64_t)co * Cin + ci) * 27 + combo * 3 + dxi] = (float)v;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           float* __restrict__ db, int Cin, int Cout, int gx, int gy,
                                                           int n_ci_tiles, int cit, int ng) {
  __shared__ double sm[3][64];
  wgrad_reduce_body<MODE>(part, dw, db, Cin, Cout, gx, gy, n_ci_tiles, cit, ng, blockIdx.x, sm);
}

// Deferred form (modet_conv3d_bwd_weight_defer + modet_conv3d_wgrad_defer_flush): the reductions of ALL weight-gradient launches of a backward
// pass in one launch.  A step has 20 of them, 5-9 us each as separate launches of 8-560 workgroups -- almost all of it
// launch + drain latency.  The job table travels by value in the kernel arguments (no device table to keep alive,
// capturable); workgroup -> job by a scalar search over the block prefix.
constexpr int REDUCE_MAX_JOBS = 32;
struct ReduceTable {
  ReduceJob job[REDUCE_MAX_JOBS];
  int first[REDUCE_MAX_JOBS + 1];
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(const ReduceTable t) {
  __shared__ double sm[3][64];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  const ReduceJob& J = t.job[j];
  const int blk = blockIdx.x - t.first[j];
  if (J.mode == 0) wgrad_reduce_body<0>(J.part, J.dw, J.db, J.Cin, J.Cout, J.gx, J.gy, J.n_ci, J.cit, J.ng, blk, sm);
  else wgrad_reduce_body<1>(J.part, J.dw, J.db, J.Cin, J.Cout, J.gx, J.gy, J.n_ci, J.cit, J.ng, blk, sm);
}

// ------------------------------------------------------------------------------------------------ wgrad, N-packed
// Cout <= 8 and Cin <= 8 (the full-resolution layers): the 16 N columns are (q, co) with q in {0,1}; column block
// q = 1 multiplies by d_y shifted one voxel in +x:
//   D[(t,ci)][(q,co)] = sum_v x[v + off(t)][ci] * dy[v + q*ex][co] = sum_u x[u + off(t) - q*ex][ci] * dy[u][co],
// i.e. the weight gradient of tap off(t) - q*ex.  With A taps dx in {0,+1} per (dz,dy) group, (t,q) = (0,1),(0,0),(1,0)
// are dx = -1, 0, +1: 9 groups (+1 bias group) instead of 14 tap pairs -> 1.5x fewer MFMAs.  The voxel set of the
// q = 1 columns is the tile shifted by +1 in x; the column it misses (u.x = 0, tap dx = -1) multiplies the
// zero-padded x[-1], so nothing is lost and no correction pass is needed.
// CIT = 8: M rows = (t in {0,1}) x 8 ci, group g = (dz,dy).   CIT = 4: M rows = (pair pp in {0,1}, t in {0,1}) x 4 ci,
// group g = two consecutive (dz,dy) combos.
constexpr int NP_DX = TX + 1;                          // d_y tile keeps one extra voxel column per row
// ROWLD (Cin == Cout == 8): the prefetch is issued as whole halo rows -- a wave owns rows w, w+4, ..., lane l loads the
// l-th float4 of the row -- so a load's address is a wave-uniform row base (scalar unit) plus a per-lane constant and its
// bounds test one uniform row test and one per-tile lane mask: almost no VALU, where the per-element form spent 35-44 %
// of a wave's cycles (tools/exp_conv_phases.py) issuing the next tile behind the other waves' MFMAs.
template <int CIT, int WG_TZ, bool ROWLD>
__global__ __launch_bounds__(NTHR) void conv3d_wgrad_np_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ part, int D, int H, int W, int Cin,
                                                               int Cout, int tiles_x, int tiles_y, int tiles_z,
                                                               int ntiles) {
  constexpr int WG_ROWS = WG_TZ * WG_TY, WG_HVOX = (WG_TZ + 2) * WG_HY * HX;
  constexpr int NCOMBO = 9, CPG = 8 / CIT;             // (dz,dy) combos per group: 1 (CIT 8) or 2 (CIT 4)
  constexpr int NGT = (NCOMBO + CPG - 1) / CPG;        // tap groups: 9 or 5
  constexpr int NG = NGT + 1;                          // + bias group
  constexpr int GPW = (NG + 3) / 4;
  __shared__ __attribute__((aligned(16))) float xs[WG_HVOX * CIT];
  __shared__ __attribute__((aligned(16))) float dys[WG_ROWS * NP_DX * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;

  f32x4 acc[GPW];
  int aoff[GPW];
  // as in conv3d_wgrad_kernel: no select in front of the MFMAs except for the one g that can hold the bias group
  // (grp == NGT: A = 1); slots past it accumulate tile data nobody reads
  constexpr int BIAS_W = NGT % 4, BIAS_G = NGT / 4;
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int grp = wave + 4 * g;                      // groups interleaved over the waves: (3,3,2,2) / (2,2,1,1)
    const int t = li / CIT, ci = li % CIT;
    const int combo = grp * CPG + (CPG == 2 ? (t >> 1) : 0), dxs = t & 1;
    const int cc = combo < NCOMBO ? combo : 0;
    aoff[g] = (((cc / 3) * WG_HY + (cc % 3)) * HX + 1 + dxs + lk) * CIT + ci;
  }
  const bool bsel = wave == BIAS_W;
  const int bq = li >> 3, bco = li & 7;                // B column = (q, co)

  constexpr int QX = CIT / 4, NXV = (WG_HVOX * QX + NTHR - 1) / NTHR;
  constexpr int NDV = (WG_ROWS * NP_DX * 2 + NTHR - 1) / NTHR;
  float4 xr[NXV], dr[NDV];
  const bool xvec = (Cin & 3) == 0, dvec = (Cout & 3) == 0;

  auto load_tile = [&](int tl) {
    int t = tl;
    const int x0 = (t % tiles_x) * TX; t /= tiles_x;
    const int y0 = (t % tiles_y) * WG_TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * WG_TZ;
    const int64_t vb = (int64_t)(t / tiles_z) * D * H * W;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int idx = tid + i * NTHR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < WG_HVOX * QX) {
        const int hv = idx / QX, c4 = idx - hv * QX;
        const int hx = hv % HX, t2 = hv / HX;
        const int hy = t2 % WG_HY, hz = t2 / WG_HY;
        const int z = z0 + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
        const int c = c4 * 4;
        if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W && c < Cin) {
          const float* p = x + (vb + ((int64_t)z * H + yy) * W + xx) * Cin + c;
          if (xvec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (c + 1 < Cin) v.y = p[1];
            if (c + 2 < Cin) v.z = p[2];
            if (c + 3 < Cin) v.w = p[3];
          }
        }
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int idx = tid + i * NTHR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < WG_ROWS * NP_DX * 2) {
        const int vox = idx >> 1, c4 = idx & 1;
        const int row = vox / NP_DX, xx = x0 + vox % NP_DX;
        const int z = z0 + row / WG_TY, yy = y0 + row % WG_TY;
        const int co = c4 * 4;
        if (z < D && yy < H && xx < W && co < Cout) {
          const float* p = dy + (vb + ((int64_t)z * H + yy) * W + xx) * Cout + co;
          if (dvec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (co + 1 < Cout) v.y = p[1];
            if (co + 2 < Cout) v.z = p[2];
            if (co + 3 < Cout) v.w = p[3];
          }
        }
      }
      dr[i] = v;
    }
  };

  // ---- row-wise prefetch (ROWLD)
  constexpr int XROWS = (WG_TZ + 2) * WG_HY, XRW = XROWS / 4;          // halo rows of x, rows per wave (60 / 4)
  constexpr int XQ = HX * 2, DQ = NP_DX * 2;                            // float4 per x row (36) / d_y row (34)
  constexpr int DRW = WG_ROWS / 4;                                      // d_y rows per wave
  static_assert(!ROWLD || (CIT == 8 && XROWS % 4 == 0 && WG_ROWS % 4 == 0), "row prefetch shape");
  float4 xq[ROWLD ? XRW : 1], dq[ROWLD ? DRW : 1];
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto load_rows = [&](int tl) {
    int t = tl;
    const int x0 = (t % tiles_x) * TX; t /= tiles_x;
    const int y0 = (t % tiles_y) * WG_TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * WG_TZ;
    const int bz = (t / tiles_z) * D;
    const int xx = x0 - 1 + (lane >> 1);
    const bool xl_ok = lane < XQ && xx >= 0 && xx < W;
    const bool dl_ok = lane < DQ && x0 + (lane >> 1) < W;
#pragma unroll
    for (int i = 0; i < XRW; ++i) {
      const int r = wv + 4 * i, hz = r / WG_HY, hy = r - hz * WG_HY;
      const int z = z0 + hz - 1, yy = y0 + hy - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z >= 0 && z < D && yy >= 0 && yy < H) {                        // wave-uniform
        const float* rp = x + (((int64_t)(bz + z) * H + yy) * W + (x0 - 1)) * 8;
        if (xl_ok) v = *reinterpret_cast<const float4*>(rp + lane * 4);
      }
      xq[i] = v;
    }
#pragma unroll
    for (int i = 0; i < DRW; ++i) {
      const int r = wv + 4 * i, rz = r / WG_TY, ry = r - rz * WG_TY;
      const int z = z0 + rz, yy = y0 + ry;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z < D && yy < H) {
        const float* rp = dy + (((int64_t)(bz + z) * H + yy) * W + x0) * 8;
        if (dl_ok) v = *reinterpret_cast<const float4*>(rp + lane * 4);
      }
      dq[i] = v;
    }
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int i = 0; i < XRW; ++i)
      if (lane < XQ) *reinterpret_cast<float4*>(xs + (wv + 4 * i) * (HX * 8) + lane * 4) = xq[i];
#pragma unroll
    for (int i = 0; i < DRW; ++i)
      if (lane < DQ) *reinterpret_cast<float4*>(dys + (wv + 4 * i) * (NP_DX * 8) + lane * 4) = dq[i];
  };

  int tile = blockIdx.x;
  if (tile < ntiles) { if (ROWLD) load_rows(tile); else load_tile(tile); }
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    if (ROWLD) {
      store_rows();
    } else {
#pragma unroll
      for (int i = 0; i < NXV; ++i) {
        const int idx = tid + i * NTHR;
        if (idx < WG_HVOX * QX) *reinterpret_cast<float4*>(xs + idx * 4) = xr[i];
      }
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int idx = tid + i * NTHR;
        if (idx < WG_ROWS * NP_DX * 2) *reinterpret_cast<float4*>(dys + idx * 4) = dr[i];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) { if (ROWLD) load_rows(tile + gridDim.x); else load_tile(tile + gridDim.x); }
    __builtin_amdgcn_s_setprio(1);
    {
      // explicit two-deep software pipeline over the (row, s) steps: the fragments of step i+1 are read from LDS
      // before the MFMAs of step i issue (-8 % on the 8->8 layer; the same rewrite made the generic kernel slower,
      // hipcc's own schedule of its fully unrolled rows is better there)
      constexpr int NSTEP = WG_ROWS * 4;
      float af[2][GPW], bfv[2];
      auto frag = [&](int st, float (&a)[GPW], float& b) {
        const int row = st >> 2, sq = st & 3;
        const int rb = ((row / WG_TY) * WG_HY + (row % WG_TY)) * HX * CIT;
        b = dys[(row * NP_DX + sq * 4 + lk + bq) * 8 + bco];
#pragma unroll
        for (int g = 0; g < GPW; ++g) a[g] = xs[rb + sq * 4 * CIT + aoff[g]];
      };
      frag(0, af[0], bfv[0]);
#pragma unroll 32
      for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) frag(st + 1, af[(st + 1) & 1], bfv[(st + 1) & 1]);
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
          float a = af[st & 1][g];
          if (g == BIAS_G) a = bsel ? 1.f : a;
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfv[st & 1], acc[g], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    const int grp = wave + 4 * g;
    if (grp >= NG) continue;
    float* p = part + ((int64_t)blockIdx.x * NG + grp) * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) p[(lk * 4 + j) * 16 + li] = acc[g][j];
  }
}

// ------------------------------------------------------------------------------------------------ Cin == 1
// First encoder conv (1 -> 4): K = 27 is far below an MFMA tile and the op is a pure HBM-bound stencil
// (reads 4 B, writes 16 B per voxel), so it runs on the VALU: one thread per voxel, all CO outputs in registers.
template <int CO>
__global__ __launch_bounds__(NTHR) void conv_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int D,
                                                           int H, int W, int64_t total, int act) {
  __shared__ float wsm[27 * CO + CO];
  for (int i = threadIdx.x; i < 27 * CO; i += NTHR) wsm[(i % 27) * CO + i / 27] = w[i];     // (CO,1,27) -> [tap][co]
  if (threadIdx.x < CO) wsm[27 * CO + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int64_t V = (int64_t)D * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * NTHR + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NTHR) {
    const int64_t b = idx / V, v = idx - b * V;
    const int xi = (int)(v % W);
    const int64_t t2 = v / W;
    const int yi = (int)(t2 % H), zi = (int)(t2 / H);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = wsm[27 * CO + c];
#pragma unroll 1
    for (int dz = 0; dz < 3; ++dz)             // not unrolled: keeps the 27*CO weights in LDS instead of VGPRs
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int zz = zi + dz - 1, yy = yi + dy - 1, xx = xi + dx - 1;
          float xv = 0.f;
          if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
            xv = x[b * V + ((int64_t)zz * H + yy) * W + xx];
          const int tap = (dz * 3 + dy) * 3 + dx;
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[c] = fmaf(xv, wsm[tap * CO + c], acc[c]);
        }
    if (act) {
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = lrelu(acc[c]);
    }
#pragma unroll
    for (int c = 0; c < CO; c += 4)
      *reinterpret_cast<float4*>(y + idx * CO + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  }
}

// The same layer as a z-MARCH for full-resolution volumes: a workgroup owns an 8 x 32 column of (y, x), one thread per voxel.
// Only the incoming plane goes through LDS (double-buffered, for the 3x3 neighbour exchange); a thread keeps the 3x3 (dy, dx)
// neighbourhoods of the previous two planes in registers, so a voxel costs 9 LDS reads instead of 27 cached global loads, the
// halo is 1.33x (y, x only) and the next plane is in flight (buffer loads: zero padding by the descriptor) while this one
// computes.  (The per-voxel kernel above: 0.110 ms at 2 x 160x192x160 = 1.8 TB/s of the 197 MB it moves.)
constexpr int C1Y = 8, C1X = 32, C1HY = C1Y + 2, C1HX = C1X + 2, C1HV = C1HY * C1HX;
struct C1Args { const float* x; const float* w; const float* bias; float* y; int D, H, W, tiles_x, tiles_y, ZC, act;
                unsigned* amax; };   // amax != null: max |y| is left there (MODET_AMAX_SLOTS slots, zeroed by the launcher)
__global__ __launch_bounds__(NTHR) void conv_c1_march_kernel(const C1Args a) {
  __shared__ float pl[2][C1HV];
  const int tid = threadIdx.x;
  const int D = a.D, H = a.H, W = a.W;
  int t = blockIdx.x;
  const int x0 = (t % a.tiles_x) * C1X; t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * C1Y;
  const int zs = (t / a.tiles_y) * a.ZC;
  const int ze = zs + a.ZC < D ? zs + a.ZC : D;
  const int b = blockIdx.y;
  const float* xb = a.x + (int64_t)b * D * H * W;
  float* yb = a.y + (int64_t)b * D * H * W * 4;
  // the four couts travel as two float PAIRS: 54 v_pk_fma_f32 per voxel instead of 108 v_fma_f32 (the kernel was VALU-bound)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 wv[27][2], bv[2];
  {
    // (4, 1, 27): wave-uniform reads, pinned into VECTOR registers (left to itself the compiler keeps 108 SGPRs and spills
    // them).  All 108 loads are issued before the first pin: pinned one by one, each load was followed by its own
    // s_waitcnt vmcnt(0) -- 108 memory round trips in series at the head of every workgroup.
    float tw[108];
#pragma unroll
    for (int i = 0; i < 108; ++i) tw[i] = a.w[i];
#pragma unroll
    for (int i = 0; i < 108; ++i) asm volatile("" : "+v"(tw[i]));
#pragma unroll
    for (int tp = 0; tp < 27; ++tp)
#pragma unroll
      for (int c = 0; c < 4; ++c) wv[tp][c >> 1][c & 1] = tw[c * 27 + tp];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) bv[c >> 1][c & 1] = a.bias ? a.bias[c] : 0.f;
  // staging: halo voxel v = tid (and 256 + tid for the first C1HV - 256 threads); byte offsets inside a plane
  using Buf = __amdgpu_buffer_rsrc_t;
  auto rsrc = [](const float* base, unsigned bytes) -> Buf {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)p);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  unsigned go[2];
  int so[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int v = tid + j * NTHR;
    const bool on = v < C1HV;
    const int hy = on ? v / C1HX : 0, hx = on ? v - hy * C1HX : 0;
    const int yy = y0 + hy - 1, xx = x0 + hx - 1;
    go[j] = (on && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)((yy * W + xx) * 4) : 0x80000000u;
    so[j] = on ? v : -1;
  }
  const unsigned plane_bytes = (unsigned)H * W * 4;
  float pr[2];
  auto load_plane = [&](int z) {
    const bool live = z >= 0 && z < D;
    const Buf rs = rsrc(xb + (int64_t)(live ? z : 0) * H * W, live ? plane_bytes : 0u);
    pr[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)go[0], 0, 0));
    pr[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)go[1], 0, 0));
  };
  auto store_plane = [&](int slot) {
    pl[slot][tid] = pr[0];
    if (so[1] >= 0) pl[slot][so[1]] = pr[1];
  };
  const int tx = tid % C1X, ty = tid / C1X;
  const int hc = ty * C1HX + tx;                           // halo index of the (-1, -1) neighbour
  auto nb9 = [&](int slot, float (&n)[9]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) n[dy * 3 + dx] = pl[slot][hc + dy * C1HX + dx];
  };
  const bool live = y0 + ty < H && x0 + tx < W;
  float n0[9], n1[9], n2[9];
  float ymax = 0.f;
  load_plane(zs - 1); store_plane(0);
  load_plane(zs);
  __syncthreads();
  nb9(0, n0);
  store_plane(1);
  load_plane(zs + 1);
  __syncthreads();
  nb9(1, n1);
  for (int z = zs, i = 0; z < ze; ++z, ++i) {
    store_plane(i & 1);                                    // plane z+1 (the slot plane z-1 went through two barriers ago)
    __syncthreads();
    load_plane(z + 2);
    nb9(i & 1, n2);
    f2 ac[2] = {bv[0], bv[1]};
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ac[h] = __builtin_elementwise_fma((f2){n0[k], n0[k]}, wv[k][h], ac[h]);
        ac[h] = __builtin_elementwise_fma((f2){n1[k], n1[k]}, wv[9 + k][h], ac[h]);
        ac[h] = __builtin_elementwise_fma((f2){n2[k], n2[k]}, wv[18 + k][h], ac[h]);
      }
    float acc[4] = {ac[0][0], ac[0][1], ac[1][0], ac[1][1]};
    if (a.act) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = lrelu(acc[c]);
    }
    if (live) {
      *reinterpret_cast<float4*>(yb + (((int64_t)z * H + y0 + ty) * W + x0 + tx) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) { n0[k] = n1[k]; n1[k] = n2[k]; }
  }
  if (a.amax) {               // (uniform) one integer atomic max per wave: non-negative floats order like their bit patterns
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
    if ((tid & 63) == 0 && ymax > 0.f) atomicMax(a.amax + (blockIdx.x % MODET_AMAX_SLOTS) * MODET_AMAX_STRIDE, __float_as_uint(ymax));
  }
}

constexpr int C1_MFMA_BLOCKS = 1024;

// Weight gradient of the first conv (Cin = 1, Cout = 4) on the matrix pipe: M = 16 taps per MFMA (two groups cover the
// 27 taps + the bias slot), N = cout (columns 4..15 repeat 0..3 and are never read), K = 4 voxels.  (A VALU version
// with 112 accumulators per thread -- 192 VGPRs, 2 waves/SIMD -- ran at 0.26-0.29 ms for 196 MB of traffic; this: 0.17.)
// Tile 4x8x16 voxels; wave w owns tile rows 8w..8w+7 with both tap groups; per-wave partial tiles go through
// wgrad_reduce_kernel<0> (cit = 1).  yact != null folds LeakyReLU' into the d_y staging (ConvBlock backward).
constexpr int C1_TZ = 4, C1_ROWS = C1_TZ * WG_TY, C1_HVOX = (C1_TZ + 2) * WG_HY * HX;
__global__ __launch_bounds__(NTHR) void conv_c1_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ yact, float* __restrict__ part,
                                                                  int D, int H, int W, int tiles_x, int tiles_y,
                                                                  int tiles_z, int ntiles) {
  __shared__ float xs[C1_HVOX + 64];
  __shared__ __attribute__((aligned(16))) float dys[C1_ROWS * TX * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  int aoff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int tap = g * 16 + li, tt = tap < 27 ? tap : 0;
    aoff[g] = ((tt / 9) * WG_HY + (tt / 3) % 3) * HX + tt % 3 + lk;
  }
  const bool bsel = li == 11;                          // group 1, row 11 = tap 27: the bias slot (A = 1)
  for (int i = tid; i < 64; i += NTHR) xs[C1_HVOX + i] = 0.f;
  // register prefetch of the next tile (x: C1_HVOX scalars, d_y: one float4 per voxel) under the MFMA loop
  constexpr int NXS = (C1_HVOX + NTHR - 1) / NTHR, NDS = C1_ROWS * TX / NTHR;
  float xr[NXS];
  bool xin[NXS];
  float4 dr[NDS], yr[NDS];
  bool din[NDS];
  auto load_tile = [&](int tl) {
    int t = tl;
    const int x0 = (t % tiles_x) * TX; t /= tiles_x;
    const int y0 = (t % tiles_y) * WG_TY; t /= tiles_y;
    const int z0 = (t % tiles_z) * C1_TZ;
    const int64_t vb = (int64_t)(t / tiles_z) * D * H * W;
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int i = tid + k * NTHR;
      const int hx = i % HX, r = i / HX;
      const int z = z0 + r / WG_HY - 1, yy = y0 + r % WG_HY - 1, xx = x0 + hx - 1;
      xin[k] = i < C1_HVOX && z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
      xr[k] = x[xin[k] ? vb + ((int64_t)z * H + yy) * W + xx : 0];          // raw; zero padding applied at the LDS write
    }
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      const int i = tid + k * NTHR;
      const int vx = i % TX, row = i / TX;
      const int z = z0 + row / WG_TY, yy = y0 + row % WG_TY, xx = x0 + vx;
      // raw loads only (clamped address, the predicate is applied when the tile is written to LDS): forming
      // d_y * LeakyReLU'(y) here made the PREFETCH wait for both loads in front of the MFMA loop it should hide under
      const bool in = z < D && yy < H && xx < W;
      const int64_t o = in ? (vb + ((int64_t)z * H + yy) * W + xx) * 4 : 0;
      dr[k] = *reinterpret_cast<const float4*>(dy + o);
      if (yact) yr[k] = *reinterpret_cast<const float4*>(yact + o);          // (uniform)
      din[k] = in;
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int i = tid + k * NTHR;
      if (i < C1_HVOX) xs[i] = xin[k] ? xr[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      float4 gv = dr[k];
      if (yact) {
        gv.x *= yr[k].x > 0.f ? 1.f : LRELU_SLOPE; gv.y *= yr[k].y > 0.f ? 1.f : LRELU_SLOPE;
        gv.z *= yr[k].z > 0.f ? 1.f : LRELU_SLOPE; gv.w *= yr[k].w > 0.f ? 1.f : LRELU_SLOPE;
      }
      if (!din[k]) gv = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(dys + (tid + k * NTHR) * 4) = gv;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rr = 0; rr < C1_ROWS / 4; ++rr) {
      const int row = wave * (C1_ROWS / 4) + rr;
      const int rb = ((row / WG_TY) * WG_HY + (row % WG_TY)) * HX;
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) {
        const float bf = dys[(row * TX + sq * 4 + lk) * 4 + (li & 3)];
        const float a0 = xs[rb + sq * 4 + aoff[0]];
        float a1 = xs[rb + sq * 4 + aoff[1]];
        a1 = bsel ? 1.f : a1;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf, acc[1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  // The four waves are summed through LDS in fixed order (deterministic), ONE partial per workgroup:
  // part[blk*2 + grp][tap row][col].  (Per-wave partials made 4 096 rows of 512 floats: the reduction -- eight quarter-tile
  // workgroups walking them in dependent batches -- took 49 us of every step for 108 weights.)
  float* red = dys;                                  // 2 x 256 floats
  for (int w4 = 0; w4 < 4; ++w4) {
    __syncthreads();
    if (wave == w4) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* slot = red + g * 256 + (lk * 4 + j) * 16 + li;
          *slot = w4 == 0 ? acc[g][j] : *slot + acc[g][j];
        }
    }
  }
  __syncthreads();
  float* pp = part + (int64_t)blockIdx.x * 512;
  for (int i = tid; i < 512; i += NTHR) pp[i] = red[i];
}

// ------------------------------------------------------------------------------------------------ host side
struct FwdPlan {
  int cfg;       // 0:A 1:B 2:C 3:D 4:A with CK=4
  int ncb, ck, tz, ty;
  int P;         // output rows packed into the N tile (cfg 0/4 only)
};
inline FwdPlan plan_fwd(int64_t BV, int Cin, int Cout) {
  const int P = Cout <= 4 ? 4 : (Cout <= 8 ? 2 : 1);
#ifdef MODET_TUNING
  if (const char* e = getenv("MODET_CONV_CFG")) {
    switch (atoi(e)) {
      case 0: if (Cout <= 16) return {0, 16, 8, 4, 8, P}; break;
      case 1: return {1, 32, 4, 4, 8, 1};
      case 2: return {2, 64, 4, 2, 4, 1};
      case 3: return {3, 64, 4, 1, 4, 1};
      case 4: if (Cout <= 16) return {4, 16, 4, 4, 8, P}; break;
      case 5: return {5, 32, 4, 2, 4, 1};
      case 6: return {6, 32, 8, 1, 4, 1};
      case 7: return {7, 64, 8, 1, 4, 1};
      case 8: return {8, 16, 4, 1, 4, 1};
      default: break;
    }
  }
#endif
  // small volumes (pyramid levels 3-5, measured with tools/sweep_conv.py): parallelism first -- 4-row tiles, narrow N
  // blocks, 8-channel stages (half the barriers)
  if (BV < 60000) {
    if (Cout <= 16) return {8, 16, 4, 1, 4, 1};
    if (Cout % 64 == 0 && BV >= 8000) return {7, 64, 8, 1, 4, 1};
    return {6, 32, 8, 1, 4, 1};
  }
  if (Cout <= 16 && Cin <= 4) return {4, 16, 4, 4, 8, P};
  if (Cout <= 16) return {0, 16, 8, 4, 8, P};
  if (Cout <= 32 && BV >= 600000) return {1, 32, 4, 4, 8, 1};     // NCB = 32: no half-empty N tiles
  if (Cout <= 32) return {5, 32, 4, 2, 4, 1};                      // level 3: smaller tiles fill the chip
  if (Cout <= 64) return {2, 64, 4, 2, 4, 1};
  return {3, 64, 4, 1, 4, 1};
}
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// resident 256-thread workgroups per CU from the kernel's own footprint: 160 KiB LDS per CU, 512 VGPR+AGPR per SIMD
// lane allocated in granules of 8 (MI355X_MICROARCH.md), at most 8 waves per SIMD
// The answer is a constant of the code object: it is queried once per kernel and memoised (a read-mostly table behind a
// mutex -- the only process-wide state of the library, immutable once filled), so that the hundreds of launches of a
// step make no runtime queries and a step can be captured into a hipGraph (no non-stream API calls while capturing).
inline int resident_blocks(const void* fn, int nthreads) {
#ifdef MODET_TUNING
  if (const char* e = getenv("MODET_CONV_PERCU")) return atoi(e);
#endif
  static std::mutex mu;
  static std::unordered_map<const void*, int> memo;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(fn);
    if (it != memo.end()) return it->second;
  }
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, fn) != hipSuccess) return 2;
  const int by_lds = a.sharedSizeBytes > 0 ? (int)(163840 / round_up((int)a.sharedSizeBytes, 512)) : 8;
  const int waves_per_simd = a.numRegs > 0 ? 512 / round_up(a.numRegs, 8) : 8;
  const int by_reg = (waves_per_simd > 8 ? 8 : waves_per_simd) * 4 / (nthreads / 64);
  int n = by_lds < by_reg ? by_lds : by_reg;
  if (n > 8) n = 8;
  n = n < 1 ? 1 : n;
  std::lock_guard<std::mutex> lk(mu);
  memo[fn] = n;
  return n;
}

inline size_t fwd_ws_elems(int Cin, int Cout) {
  // generous: any plan pads Cin to <= 8 and Cout to <= 64 granules; row packing uses up to 54 taps x 16 columns
  const size_t plain = (size_t)27 * round_up(Cin, 16) * round_up(Cout, 64);       // (16: conv_direct_kernel's channel blocks)
  const size_t packed = (size_t)54 * round_up(Cin, 8) * 16;
  return plain > packed ? plain : packed;
}

// ---- weight packing hoisted out of the step (modet_conv3d_prepack_*, state in the caller's modet_step_ctx).  While
// RECORDING, every conv launch that is given the context notes its packing job (weights pointer, mode, padded geometry);
// _begin packs all recorded jobs in one launch into the caller's arena and, until _end, a conv launch whose job is in the
// table uses the arena copy and skips its own packing launch.
inline size_t pack_job_elems(const PackKey& k) { return ((size_t)9 * (k.P + 2) * k.CinP * k.CoutP + 63) / 64 * 64; }

// returns the pre-packed weights for this launch, or null (and records the job when the context is recording)
static const float* prepacked_or_record(modet_step_ctx* c, const PackKey& k) {
  if (!c) return nullptr;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->active) {
    for (size_t i = 0; i < c->jobs.size(); ++i)
      if (c->jobs[i] == k) return c->arena + c->off[i];
    return nullptr;
  }
  if (c->recording) {
    bool seen = false;
    for (const PackKey& j : c->jobs) seen = seen || j == k;
    if (!seen) c->jobs.push_back(k);
  }
  return nullptr;
}

// ------------------------------------------------------------------------------------------------ small volumes
// Forward / dgrad of the coarse levels (<= 16 k voxels: encoder level 5, the CWM layers at level-4 resolution).  The tiled
// kernel above stages 1x4x16-voxel tiles through LDS: at 10x12x10 a row of 16 holds 10 voxels, ~240 workgroups of 4 waves
// serialise 16-32 stages behind two barriers each, and the launch runs at 24-40 TFLOP/s.  Here the whole problem sits in L2
// (1-5 MB of activations, < 2 MB of weights), so nothing is staged at all: ONE WAVE owns 16 consecutive voxels of the flattened
// volume (no ragged rows) x 16*NT output channels (its four waves: a quarter of the 27 taps each), and feeds v_mfma_f32_16x16x4_f32 (exact
// fp32) straight from global memory: lane (voxel i, k) loads float4 = channels c0 + 4k .. c0 + 4k + 3 of its voxel's tap
// neighbour (clamped address, zero by select outside the volume), lane (k, n) the matching float4 of the direct weight layout
// (pack_direct); MFMA j of the four takes component j of both.  U channel blocks are loaded and fenced together; no LDS, no
// barrier in the main loop, no tile geometry.
template <int NT, int U>
__global__ __launch_bounds__(256) void conv_direct_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                          const float* __restrict__ bias, float* __restrict__ y, int D, int H,
                                                          int W, int Cin, int Cout, int CS, int CT, int BV) {
  // the four waves of a workgroup split the 27 taps of ONE (voxel tile, output tile) -- K split four ways, partial tiles
  // summed through LDS in fixed order: a single wave per tile is a chain of ~1 us load round trips with 0.25 us of MFMAs
  // between them (measured: 50 us where the matrix pipe needs 15), four to five waves per SIMD hide it
  __shared__ float red[3][NT][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int vt = blockIdx.x;
  const int ct0 = blockIdx.y * NT;
  const int v = vt * 16 + li < BV ? vt * 16 + li : BV - 1;   // rows past the end repeat the last voxel (never stored)
  const unsigned uv = (unsigned)v, q1 = uv / (unsigned)W, q2 = q1 / (unsigned)H;
  const int xi = (int)(uv - q1 * (unsigned)W), yi = (int)(q1 - q2 * (unsigned)H), zi = (int)(q2 % (unsigned)D);
  const float* xv = x + (int64_t)v * Cin;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // trips = (tap of this wave, U channel blocks); the loads of trip i + 1 are issued before the MFMAs of trip i (two register
  // sets), so a wave's chain costs max(load latency, MFMA time) per trip instead of their sum
  const int TPT = CS / U;                                    // trips per tap (CS % U == 0 by dispatch)
  const int ntrip = ((27 - wave + 3) / 4) * TPT;
  auto issue = [&](int trip, float4 (&a)[U], float4 (&b)[U][NT], bool& inb) {
    const int tap = wave + 4 * (trip / TPT), cs0 = (trip % TPT) * U;
    const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    inb = zi + dz >= 0 && zi + dz < D && yi + dy >= 0 && yi + dy < H && xi + dx >= 0 && xi + dx < W;
    const float* xa = inb ? xv + ((int64_t)(dz * H + dy) * W + dx) * Cin : xv;
    const float* wb = wpk + (int64_t)tap * CS * CT * 256 + lane * 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int coff = (cs0 + u) * 16 + 4 * lk;                    // a channel block past Cin meets zero weights: any valid address
      coff = coff <= Cin - 4 ? coff : Cin - 4;
      a[u] = *reinterpret_cast<const float4*>(xa + coff);
#pragma unroll
      for (int t = 0; t < NT; ++t) b[u][t] = *reinterpret_cast<const float4*>(wb + (int64_t)((cs0 + u) * CT + ct0 + t) * 256);
    }
  };
  auto consume = [&](float4 (&a)[U], float4 (&b)[U][NT], bool inb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {                            // this trip's loads have landed (all of them, at once)
      asm volatile("" : "+v"(a[u].x), "+v"(a[u].y), "+v"(a[u].z), "+v"(a[u].w));
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(b[u][t].x));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!inb) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u][t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u][t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u][t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u][t].w, acc[t], 0, 0, 0);
      }
    }
  };
  float4 a0[U], b0[U][NT], a1[U], b1[U][NT];
  bool in0 = false, in1 = false;
  // branch-free body (a trip index past the end re-loads the last trip and is consumed with zeros): at a control-flow join
  // the compiler's wait insertion falls back to s_waitcnt vmcnt(0), which would wait for the prefetch it just issued
  const int last = ntrip - 1;                                // ntrip >= 6
  issue(0, a0, b0, in0);
  for (int trip = 0; trip < ntrip; trip += 2) {
    issue(trip + 1 < ntrip ? trip + 1 : last, a1, b1, in1);
    consume(a0, b0, in0);
    issue(trip + 2 < ntrip ? trip + 2 : last, a0, b0, in0);
    consume(a1, b1, in1 && trip + 1 < ntrip);
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave - 1][t][j * 64 + lane] = acc[t][j];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = (ct0 + t) * 16 + li;
    if (co >= Cout) continue;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = vt * 16 + lk * 4 + j;
      const float sum = ((acc[t][j] + red[0][t][j * 64 + lane]) + red[1][t][j * 64 + lane]) + red[2][t][j * 64 + lane];
      if (row < BV) y[(int64_t)row * Cout + co] = sum + bv;
    }
  }
}

// MODET_CONV_DIRECT=0 keeps the tiled kernels for every shape (A/B switch)
static bool use_direct(int B, int D, int H, int W, int Cin, int Cout) {
  static const bool on = modet_tuning_env("MODET_CONV_DIRECT") != '0';
  const int64_t BV = (int64_t)B * D * H * W;
  // (up to 100 k voxels -- the CWM layers at level-3 resolution -- measured: 9.655 vs 9.590 ms/step; the operands then stream from HBM)
  return on && BV <= 16384 && Cin >= 8 && Cin % 4 == 0 && Cout >= 4;
}

static void conv_direct_launch(const float* x, const float* wpk, const float* bias, float* y, int B, int D, int H, int W,
                               int Cin, int Cout, hipStream_t s) {
  const int BV = B * D * H * W;
  const int CS = round_up(Cin, 16) / 16, CT = round_up(Cout, 16) / 16;
  const int VT = cdiv(BV, 16);
  // two output tiles per workgroup (half the A loads) once one tile each would put more than ~8 waves on a SIMD
  const bool nt2 = CT % 2 == 0 && (int64_t)VT * CT > 512;
  const dim3 grid(VT, nt2 ? CT / 2 : CT);
#define DIRECT_LAUNCH(NT_, U_) hipLaunchKernelGGL((conv_direct_kernel<NT_, U_>), grid, dim3(256), 0, s, x, wpk, bias, y, D, H, W, Cin, Cout, CS, CT, BV)
#define DIRECT_U(NT_)                                  \
  do {                                                 \
    if (CS % 4 == 0) DIRECT_LAUNCH(NT_, 4);            \
    else if (CS % 3 == 0) DIRECT_LAUNCH(NT_, 3);       \
    else if (CS % 2 == 0) DIRECT_LAUNCH(NT_, 2);       \
    else DIRECT_LAUNCH(NT_, 1);                        \
  } while (0)
  if (nt2) DIRECT_U(2); else DIRECT_U(1);
#undef DIRECT_U
#undef DIRECT_LAUNCH
}

// query_gx != null: only report the persistent grid's x size (the statistics layout depends on it), launch nothing
int conv_launch(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, float* wpk, int B, int D,
                int H, int W, int Cin, int Cout, int act, int pack_mode, hipStream_t s, float* stats = nullptr,
                int* query_gx = nullptr, ConvIn inorm = ConvIn{nullptr, nullptr, 0, nullptr}, bool query_xf = false,
                bool query_st = false) {
  if (!query_gx && !stats && !inorm.mean && !act && use_direct(B, D, H, W, Cin, Cout)) {
    const int CinP = round_up(Cin, 16), CoutP = round_up(Cout, 16), total = 27 * CinP * CoutP;
    const float* pre = prepacked_or_record(step, PackKey{w, Cin, Cout, CinP, CoutP, pack_mode + 2, 1});
    if (pre)
      wpk = const_cast<float*>(pre);
    else
      hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0, s, w,
                         wpk, Cin, Cout, CinP, CoutP, pack_mode + 2, 1);
    conv_direct_launch(x, wpk, bias, y, B, D, H, W, Cin, Cout, s);
    return modet_launch_status();
  }
  const FwdPlan p = plan_fwd((int64_t)B * D * H * W, Cin, Cout);
  const int CinP = round_up(Cin, p.ck), CoutP = round_up(Cout, p.ncb);
  const int total = 9 * (p.P + 2) * CinP * CoutP;
  if (!query_gx) {
    const float* pre = prepacked_or_record(step, PackKey{w, Cin, Cout, CinP, CoutP, pack_mode, p.P});
    if (pre)
      wpk = const_cast<float*>(pre);
    else
      hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0, s, w,
                         wpk, Cin, Cout, CinP, CoutP, pack_mode, p.P);
    if (stats) {
      // the statistics buffer starts with the [B][Cout] shift header; the partial rows follow it
      hipLaunchKernelGGL(conv_shift_kernel, dim3(cdiv(B * Cout, 4)), dim3(256), 0, s, x, w, bias, inorm.mean, inorm.rstd,
                         stats, B, D, H, W, Cin, Cout);
      inorm.shift = stats;
      stats += (size_t)B * Cout;
    }
  }
  const int tiles_x = cdiv(W, TX), tiles_y = cdiv(H, p.ty), tiles_z = cdiv(D, p.tz);
  const int ntiles = tiles_x * tiles_y * tiles_z * B;
  const int gy = CoutP / p.ncb;
  // persistent grid: exactly the resident workgroups (occupancy query is host-only and cheap), walking the tiles
#define CONV_LAUNCH_X(XF_, ST_, TZ_, TY_, WM_, WN_, ...)                                                           \
  do {                                                                                                            \
    constexpr int nthr = (WM_) * (WN_) * 64;                                                                      \
    const int per_cu = resident_blocks((const void*)conv3d_mfma_kernel<TZ_, TY_, WM_, WN_, __VA_ARGS__, XF_, ST_>, nthr); \
    int gx = (256 * per_cu + gy - 1) / gy;                                                                        \
    if (gx > ntiles) gx = ntiles;                                                                                 \
    if (query_gx) { *query_gx = gx; break; }                                                                      \
    hipLaunchKernelGGL((conv3d_mfma_kernel<TZ_, TY_, WM_, WN_, __VA_ARGS__, XF_, ST_>), dim3(gx, gy), dim3(nthr), 0, s, x, \
                       (const float*)wpk, bias, y, D, H, W, Cin, Cout, CinP, CoutP, act, tiles_x, tiles_y, tiles_z, \
                       ntiles, stats, inorm);                                                                            \
  } while (0)
  // XF kernels only where their features are used: a lazily normalised input, or statistics from a direct-store epilogue.
  // ST kernels: statistics from the staged epilogue (Cout 4/8/16 -> configurations 0, 4, 8 only).  Everything else
  // (dgrad, plain convs) runs the instantiation without any statistics code.
  const bool staged = Cout == 4 || Cout == 8 || Cout == 16;
  const bool xf = inorm.mean != nullptr || (stats != nullptr && !staged) || (query_gx != nullptr && query_xf);
  const bool st = !xf && staged && (stats != nullptr || (query_gx != nullptr && query_st));
#define CONV_LAUNCH(TZ_, TY_, WM_, WN_, ...)                                                                      \
  do {                                                                                                            \
    if (xf) CONV_LAUNCH_X(true, false, TZ_, TY_, WM_, WN_, __VA_ARGS__);                                          \
    else CONV_LAUNCH_X(false, false, TZ_, TY_, WM_, WN_, __VA_ARGS__);                                            \
  } while (0)
#define CONV_LAUNCH_S(TZ_, TY_, WM_, WN_, ...)                                                                    \
  do {                                                                                                            \
    if (xf) CONV_LAUNCH_X(true, false, TZ_, TY_, WM_, WN_, __VA_ARGS__);                                          \
    else if (st) CONV_LAUNCH_X(false, true, TZ_, TY_, WM_, WN_, __VA_ARGS__);                                     \
    else CONV_LAUNCH_X(false, false, TZ_, TY_, WM_, WN_, __VA_ARGS__);                                            \
  } while (0)
  const bool v4 = (Cin & 3) == 0;
  const bool multi = CinP > p.ck;
  switch (p.cfg) {
#define CONV_VM(L, PP, ...)                                                           \
    if (v4 && multi) L(__VA_ARGS__, true, PP, true);                                  \
    else if (v4) L(__VA_ARGS__, true, PP, false);                                     \
    else if (multi) L(__VA_ARGS__, false, PP, true);                                  \
    else L(__VA_ARGS__, false, PP, false)
#define CONV_CASE(...) CONV_VM(CONV_LAUNCH, 1, __VA_ARGS__)
#define CONV_CASE_S(...) CONV_VM(CONV_LAUNCH_S, 1, __VA_ARGS__)
#define CONV_CASE_P(...)                          \
    if (p.P == 4) { CONV_VM(CONV_LAUNCH_S, 4, __VA_ARGS__); }      \
    else if (p.P == 2) { CONV_VM(CONV_LAUNCH_S, 2, __VA_ARGS__); } \
    else { CONV_VM(CONV_LAUNCH_S, 1, __VA_ARGS__); }
    case 0: CONV_CASE_P(4, 8, 8, 1, 1, 8); break;
    case 1: CONV_CASE(4, 8, 4, 1, 2, 4); break;
    case 2: CONV_CASE(2, 4, 2, 2, 2, 4); break;
    case 4: CONV_CASE_P(4, 8, 8, 1, 1, 4); break;
    case 5: CONV_CASE(2, 4, 4, 1, 2, 4); break;      // 8 rows x N=32, 2 rows per wave            (level 3)
    case 6: CONV_CASE(1, 4, 2, 2, 1, 8); break;      // 4 rows x N=32, 8-channel stages           (levels 4-5)
    case 7: CONV_CASE(1, 4, 2, 4, 1, 8); break;      // 4 rows x N=64, 8 waves, 8-channel stages  (level 4, Cout % 64 == 0)
    case 8: CONV_CASE_S(1, 4, 4, 1, 1, 4); break;    // 4 rows x N=16, no row packing             (small volumes, Cout <= 16)
    default: CONV_CASE(1, 4, 1, 4, 1, 4); break;
  }
#undef CONV_CASE
#undef CONV_CASE_S
#undef CONV_CASE_P
#undef CONV_VM
#undef CONV_LAUNCH
#undef CONV_LAUNCH_S
#undef CONV_LAUNCH_X
  return query_gx ? MODET_OK : modet_launch_status();
}

// persistent grid (x) of the statistics-producing instantiations for this shape (ST and XF can differ in registers,
// hence in residency); the statistics buffer is sized by the larger one (conv_stats_rows)
inline int conv_grid_x(int B, int D, int H, int W, int Cin, int Cout, bool xf) {
  int gx = 0;
  conv_launch(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, D, H, W, Cin, Cout, 0, 0, nullptr, nullptr, &gx,
              ConvIn{nullptr, nullptr, 0, nullptr}, xf, !xf);
  return gx;
}

struct WgPlan { int cit, n_ci, n_co, gx, gy, ng, ntiles, tiles_x, tiles_y, tiles_z, tz; bool np; };
// persistent grid = the resident workgroups (all tiles in ONE round, and as few partial d_w copies as possible)
inline int wgrad_resident(int cit, int tz, bool np) {
  const void* fn;
  if (np) fn = tz == 4 ? (const void*)conv3d_wgrad_np_kernel<8, 4, false> : (const void*)conv3d_wgrad_np_kernel<8, 2, false>;
  else if (cit == 4) fn = tz == 4 ? (const void*)conv3d_wgrad_kernel<4, 4> : (const void*)conv3d_wgrad_kernel<4, 2>;
  else if (cit == 8) fn = tz == 4 ? (const void*)conv3d_wgrad_kernel<8, 4> : (const void*)conv3d_wgrad_kernel<8, 2>;
  else fn = (const void*)conv3d_wgrad_kernel<16, 2>;
  int r = 256 * resident_blocks(fn, NTHR);
  return r > 1024 ? 1024 : r;
}
inline WgPlan plan_wgrad(int B, int D, int H, int W, int Cin, int Cout) {
  WgPlan p;
  p.np = Cin > 4 && Cin <= 8 && Cout <= 8;   // N-packed kernel (Cin <= 4 already fills M with 4 taps: no gain there)
  p.cit = Cin <= 4 ? 4 : (Cin <= 8 ? 8 : 16);
  // channel counts that are not multiples of 16 (the CWM layers: 12, 24): the largest tile that divides Cin wastes no M rows
  // (Cin = 12 as three 4-channel tiles: 21 row tiles instead of 28 with 12 of 16 rows used; Cin = 24 as three 8-channel tiles)
  if (Cin > 8 && Cin % 16 != 0) p.cit = Cin % 8 == 0 ? 8 : (Cin % 4 == 0 ? 4 : 16);
  p.n_ci = p.np ? 1 : cdiv(Cin, p.cit);
  p.n_co = p.np ? 1 : cdiv(Cout, 16);
  p.gy = p.n_ci * p.n_co;
  p.tz = (p.cit <= 8 && (int64_t)B * D * H * W >= 1000000) ? 4 : 2;     // big tiles for the full-resolution, few-channel layers
  p.tiles_x = cdiv(W, TX); p.tiles_y = cdiv(H, WG_TY); p.tiles_z = cdiv(D, p.tz);
  p.ntiles = B * p.tiles_x * p.tiles_y * p.tiles_z;
  int gx = wgrad_resident(p.cit, p.tz, p.np) / p.gy;
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  p.gx = gx;
  p.ng = p.np ? 10 : (28 + 16 / p.cit - 1) / (16 / p.cit);
  return p;
}

}  // namespace

// fp32 emulation on the bf16 matrix pipe (conv3d_bf16.hip, "bf16x3"): fp32 tensors, six exact bf16 piece products per
// multiply, error <= 3 * 2^-24 |a b| -- the accuracy class of an fp32 FMA at 2.7x less matrix-pipe time.  OPT-IN
// (MODET_CONV_SPLIT=1) for every eligible shape (Cin % 4 == 0, Cin > 1, Cout % 4 == 0, no fused activation): all parity
// tests pass with it, but as measured in round 2 (profiles/r02d_split_vs_exact.txt) it wins 1.2-1.4x only at pyramid levels
// 2-3, ties at level 1 (those kernels are bound by staging the fp32 tile, not by the matrix pipe) and loses at levels
// 4-5, so the train step does not move (12.37 vs 12.46 ms) and the default stays the exact-f32 MFMA kernels of this file.
size_t modetx_bf16_prepack_bytes(modet_step_ctx* c);
void modetx_bf16_prepack_begin(modet_step_ctx* c, void* arena, hipStream_t stream);
void modetx_bf16_defer_flush(modet_step_ctx* c, hipStream_t stream);
bool modetx_split_eligible(int Cin, int Cout);
size_t modetx_split_ws_bytes(int Cin, int Cout);
size_t modetx_split_stats_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modetx_split_conv(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats,
                      int B, int D, int H, int W, int Cin, int Cout, int mode, hipStream_t s);
// bf16x3 z-marching kernels of the few-channel full-resolution layers (conv3d_x3.hip): the default for the shapes they
// cover; MODET_CONV_X3=0 selects the exact-f32 MFMA kernels of this file for every shape (A/B switch)
bool modetx_x3_eligible(int B, int D, int H, int W, int Cin, int Cout);
size_t modetx_x3_ws_bytes(int Cin, int Cout);
size_t modetx_x3_stats_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modetx_x3_conv(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats,
                   const float* in_mean, const float* in_rstd, int B, int D, int H, int W, int Cin, int Cout, int act, int mode,
                   hipStream_t s, const float* amax = nullptr, bool x_free = false);
size_t modetx_x3_bst_rows_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modetx_x3_dgrad_bst(modet_step_ctx* step, const float* dy, const float* w, float* dx, const float* xraw, const float* mean,
                        const float* rstd, float* rows, void* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t s,
                        const float* amax = nullptr);
bool modetx_x3_wgrad_eligible(int B, int D, int H, int W, int Cin, int Cout);
size_t modetx_x3_wgrad_ws_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modetx_x3_wgrad(modet_step_ctx* defer, const float* x, const float* dy, float* dw, float* db, void* ws, int B, int D, int H,
                    int W, int Cin, int Cout, hipStream_t s, const float* amax = nullptr, const float* in_mean = nullptr,
                    const float* in_rstd = nullptr, const float* xamax = nullptr);
// bf16x3 weight gradient through LDS transpose reads (conv3d_wtr.hip): every layer the z-march kernel does not take
// (Cin >= 12, and the few-channel layers below its voxel threshold); MODET_CONV_WTR=0 restores the exact-f32 kernels (A/B switch)
bool modetx_wtr_eligible(int B, int D, int H, int W, int Cin, int Cout);
size_t modetx_wtr_ws_bytes(int B, int D, int H, int W, int Cin, int Cout);
bool modetx_wtr_batches(int B, int D, int H, int W);
void modetx_wtr_flush(modet_step_ctx* c, hipStream_t s);
int modetx_wtr_wgrad(modet_step_ctx* defer, const float* x, const float* dy, float* dw, float* db, void* ws, int B, int D, int H,
                     int W, int Cin, int Cout, hipStream_t s, const float* amax = nullptr);
static bool use_x3(int B, int D, int H, int W, int Cin, int Cout) {
  static const bool on = modet_tuning_env("MODET_CONV_X3") != '0';
  return on && modetx_x3_eligible(B, D, H, W, Cin, Cout);
}
static bool use_x3_wgrad(int B, int D, int H, int W, int Cin, int Cout) {
  static const bool on = modet_tuning_env("MODET_CONV_X3") != '0';
  static const bool wtr_first = modet_tuning_env("MODET_CONV_WTR") == '2';      // experiment: the transpose-read kernel everywhere
  // Cout = 16 (8 -> 16 at level 2) runs 1.7x faster on the transpose-read kernel (102 -> 60 us): the march keeps Cout <= 8
  return on && !wtr_first && Cout <= 8 && modetx_x3_wgrad_eligible(B, D, H, W, Cin, Cout);
}
// bf16x3 forward / data gradient with the K index packed in channel quads (conv3d_q.hip): everything the z-march kernel does
// not take -- pyramid levels 3-5, the CWM layers, odd channel counts, launches with a lazily normalised input -- up to 1.5 M
// voxels; MODET_CONV_Q=0 (tuning builds) restores the tiled bf16x3 / exact-f32 / direct kernels
bool modetx_q_eligible(int B, int D, int H, int W, int Cin, int Cout);
size_t modetx_q_ws_bytes(int Cin, int Cout);
size_t modetx_q_stats_bytes(int B, int D, int H, int W, int Cin, int Cout);
int modetx_q_conv(modet_step_ctx* step, const float* x, const float* w, const float* bias, float* y, void* ws, float* stats,
                  const float* in_mean, const float* in_rstd, int B, int D, int H, int W, int Cin, int Cout, int mode,
                  hipStream_t s, const float* amax = nullptr, const float* xraw = nullptr, const float* bmean = nullptr,
                  const float* brstd = nullptr, float* bst_rows = nullptr, bool x_free = false);
size_t modetx_q_bst_rows_bytes(int B, int D, int H, int W, int Cin, int Cout);
static bool use_q(int B, int D, int H, int W, int Cin, int Cout) {
  static const bool on = modet_tuning_env("MODET_CONV_Q") != '0';
  // (level 5 -- 2.4 k voxels, 64 / 128 channels -- stays on conv_direct_kernel: 8 stagings of a 128-voxel tile in a row there,
  // 40-62 us against 40-55; the CWM layers at level-4 resolution, 9.6 k voxels, are faster here: 24->48 35 -> 26 us, 48->48 41 -> 33)
  const int64_t n = (int64_t)B * D * H * W;
  // (up to 6 M voxels: at cfg 5's shape -- 160x192x224, two pairs per GPU -- the CWM layers of level 3 run at 1.72 M voxels; with
  // the round-4 limit of 1.5 M they fell back to the exact-f32 kernels there, 0.42 ms of its 15 ms step)
  return on && Cin > 1 && !use_x3(B, D, H, W, Cin, Cout) && !(n < 4096 && use_direct(B, D, H, W, Cin, Cout)) && n <= 6000000 &&
         modetx_q_eligible(B, D, H, W, Cin, Cout);
}
static bool use_wtr_wgrad(int B, int D, int H, int W, int Cin, int Cout) {
  static const bool on = modet_tuning_env("MODET_CONV_WTR") != '0';
  return on && !use_x3_wgrad(B, D, H, W, Cin, Cout) && modetx_wtr_eligible(B, D, H, W, Cin, Cout);
}
// The tiled bf16x3 kernels of conv3d_bf16.hip (SP = 3): default for the MID levels of the pyramid -- Cin >= 16 channels at
// 16 k .. 1 M voxels (levels 3-4: 16->32 / 32->32 forward 0.077 / 0.131 -> 0.054 / 0.086 ms, 64->64 0.080 -> 0.067); the
// few-channel full-resolution layers take conv3d_x3.hip, level 5 (2.4 k voxels) stays on the exact-f32 kernels.
// MODET_CONV_SPLIT=1 forces them for every eligible shape, =0 switches them off.
static bool use_split(int Cin, int Cout, int64_t nvox = -1) {
  static const int mode = [] { const char e = modet_tuning_env("MODET_CONV_SPLIT"); return e ? (e == '1' ? 1 : 0) : -1; }();
  if (!modetx_split_eligible(Cin, Cout) || mode == 0) return false;
  if (mode == 1) return true;
  return nvox >= 16000 && Cin >= 16 && Cin % 16 == 0 && Cout >= 16;
}

extern "C" {

#ifdef MODET_TUNING
int modet_debug_conv_timing(long long* buf) {       // not in the header: tuning builds only
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_dbg), &buf, sizeof(buf));
}
#endif

int modet_conv3d_kernel_family_v(int B, int D, int H, int W, int Cin, int Cout, int pass, int variant) {
  if (Cin == 1) return 0;
  if (pass == 2) return use_x3_wgrad(B, D, H, W, Cin, Cout) ? 2 : (use_wtr_wgrad(B, D, H, W, Cin, Cout) ? 4 : 0);
  const int ci = pass == 1 ? Cout : Cin, co = pass == 1 ? Cin : Cout;      // the data gradient convolves d_y (Cout channels)
  if (use_x3(B, D, H, W, ci, co)) return 2;
  if (variant != 1 && use_q(B, D, H, W, ci, co)) return 5;     // (a fused activation only occurs in the 1 -> 4 ConvBlock)
  if (variant == 1 || variant == 2) return 0;                  // fused activation / normalised input: exact-f32 tiles otherwise
  if (use_split(ci, co, (int64_t)B * D * H * W)) return 1;
  if (variant == 3) return 0;                                  // fused statistics: never the direct kernel
  return use_direct(B, D, H, W, ci, co) ? 3 : 0;               // 3: conv_direct_kernel (plain forward / dgrad launches only)
}
int modet_conv3d_kernel_family(int B, int D, int H, int W, int Cin, int Cout, int pass) {
  return modet_conv3d_kernel_family_v(B, D, H, W, Cin, Cout, pass, 0);
}

int modet_step_ctx_create(modet_step_ctx_t** out) {
  MODET_CHECK_PTR(out);
  *out = new (std::nothrow) modet_step_ctx();
  return *out ? MODET_OK : MODET_ERR_WORKSPACE;
}

int modet_step_ctx_destroy(modet_step_ctx_t* ctx) {
  delete ctx;
  return MODET_OK;
}

int modet_conv3d_prepack_record(modet_step_ctx_t* c, int on) {
  MODET_CHECK_PTR(c);
  std::lock_guard<std::mutex> lk(c->mu);
  if (on) { c->jobs.clear(); c->off.clear(); c->bjobs.clear(); c->boff.clear(); c->active = false; }
  c->recording = on != 0;
  return (int)(c->jobs.size() + c->bjobs.size());
}

static size_t prepack_f32_bytes(modet_step_ctx* c) {
  std::lock_guard<std::mutex> lk(c->mu);
  size_t n = 0;
  for (const PackKey& k : c->jobs) n += pack_job_elems(k);
  return n * sizeof(float);
}

size_t modet_conv3d_prepack_arena_bytes(modet_step_ctx_t* c) {
  return c ? prepack_f32_bytes(c) + modetx_bf16_prepack_bytes(c) : 0;
}

int modet_conv3d_prepack_begin(modet_step_ctx_t* c, void* arena, size_t arena_bytes, modet_stream_t stream) {
  MODET_CHECK_PTR(c);
  std::vector<PackKey> jobs;
  std::vector<size_t> off;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->recording) return MODET_ERR_UNSUPPORTED;         // stop recording first
  }
  const size_t f32_bytes = prepack_f32_bytes(c), b16_bytes = modetx_bf16_prepack_bytes(c);
  if (f32_bytes + b16_bytes == 0) return MODET_OK;
  if (arena == nullptr) return MODET_ERR_NULL;
  if (arena_bytes < f32_bytes + b16_bytes) return MODET_ERR_WORKSPACE;
  modetx_bf16_prepack_begin(c, (char*)arena + f32_bytes, (hipStream_t)stream);      // 16-bit jobs follow the fp32 ones
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->off.assign(c->jobs.size(), 0);
    size_t n = 0;
    for (size_t i = 0; i < c->jobs.size(); ++i) { c->off[i] = n; n += pack_job_elems(c->jobs[i]); }
    c->arena = (float*)arena;
    c->active = true;
    jobs = c->jobs;                                         // the launch tables are built from copies taken under the lock
    off = c->off;
  }
  for (size_t i0 = 0; i0 < jobs.size(); i0 += PACK_MAX_JOBS) {
    PackTable t;
    const int n = (int)(jobs.size() - i0 < (size_t)PACK_MAX_JOBS ? jobs.size() - i0 : (size_t)PACK_MAX_JOBS);
    int most = 1;
    for (int i = 0; i < n; ++i) {
      const PackKey& k = jobs[i0 + i];
      t.job[i] = PackJob{k.w, (float*)arena + off[i0 + i], k.Cin, k.Cout, k.CinP, k.CoutP, k.mode, k.P, 0};
      const int blocks = cdiv(9 * (k.P + 2) * k.CinP * k.CoutP, 256);
      most = blocks > most ? blocks : most;
    }
    t.n = n;
    hipLaunchKernelGGL(pack_weights_many_kernel, dim3(most > 64 ? 64 : most, n), dim3(256), 0, (hipStream_t)stream, t);
  }
  return modet_launch_status();
}

int modet_conv3d_prepack_end(modet_step_ctx_t* c) {
  MODET_CHECK_PTR(c);
  std::lock_guard<std::mutex> lk(c->mu);
  c->active = false;
  return MODET_OK;
}

size_t modet_conv3d_ws_bytes(int Cin, int Cout) {
  const int m = Cin > Cout ? Cin : Cout;       // bwd_data swaps the roles
  size_t a = fwd_ws_elems(m, m) * sizeof(float);
  const size_t b = modetx_split_ws_bytes(Cin, Cout), c = modetx_x3_ws_bytes(Cin, Cout), d = modetx_q_ws_bytes(m, m);
  a = a > b ? a : b;
  a = a > c ? a : c;
  return a > d ? a : d;
}

// x_free: nothing is known about the range of x -> the bf16x3 forms (fp32's range) in the families that have an f16 form
// y_amax != null: max |y| is left there (only the kernel of the first encoder block carries that epilogue: UNSUPPORTED otherwise)
static int conv3d_fwd_impl(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                           int D, int H, int W, int Cin, int Cout, int act, modet_stream_t stream, modet_step_ctx_t* step, bool x_free,
                           float* y_amax = nullptr) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(w); MODET_CHECK_PTR(y); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (ws_bytes < fwd_ws_elems(Cin, Cout) * sizeof(float)) return MODET_ERR_WORKSPACE;
  const bool c1_march = Cin == 1 && Cout == 4 && (int64_t)D * H * W >= 500000 && (int64_t)H * W * 4 < 0x7fffffffLL;
  if (y_amax && !c1_march) return MODET_ERR_UNSUPPORTED;
  if (c1_march) {
    if (y_amax) modet_zero_async(y_amax, MODET_AMAX_FLOATS * sizeof(float), (hipStream_t)stream);
    const int tx = cdiv(W, C1X), ty = cdiv(H, C1Y);
    int n = cdiv(2048, B * tx * ty);                       // ~8 workgroups per CU, chunks of >= 8 planes
    const int maxn = D / 8 > 0 ? D / 8 : 1;
    n = n < 1 ? 1 : (n > maxn ? maxn : n);
    const int zc = cdiv(D, n);
    hipLaunchKernelGGL(conv_c1_march_kernel, dim3(tx * ty * cdiv(D, zc), B), dim3(NTHR), 0, (hipStream_t)stream,
                       C1Args{x, w, bias, y, D, H, W, tx, ty, zc, act, (unsigned*)y_amax});
    return modet_launch_status();
  }
  if (Cin == 1 && (Cout == 4 || Cout == 8)) {
    const int64_t total = (int64_t)B * D * H * W;
    const int grid = flat_grid(total, NTHR);
    if (Cout == 4) hipLaunchKernelGGL(conv_c1_fwd_kernel<4>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, x, w, bias, y, D, H, W, total, act);
    else hipLaunchKernelGGL(conv_c1_fwd_kernel<8>, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, x, w, bias, y, D, H, W, total, act);
    return modet_launch_status();
  }
  if (use_x3(B, D, H, W, Cin, Cout)) {
    if (ws_bytes < modetx_x3_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    return modetx_x3_conv(step, x, w, bias, y, ws, nullptr, nullptr, nullptr, B, D, H, W, Cin, Cout, act, 0, (hipStream_t)stream, nullptr, x_free);
  }
  if (!act && use_q(B, D, H, W, Cin, Cout)) {
    if (ws_bytes < modetx_q_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    return modetx_q_conv(step, x, w, bias, y, ws, nullptr, nullptr, nullptr, B, D, H, W, Cin, Cout, 0, (hipStream_t)stream, nullptr, nullptr, nullptr,
                         nullptr, nullptr, x_free);
  }
  if (!act && use_split(Cin, Cout, (int64_t)B * D * H * W)) {
    if (ws_bytes < modetx_split_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    return modetx_split_conv(step, x, w, bias, y, ws, nullptr, B, D, H, W, Cin, Cout, 0, (hipStream_t)stream);
  }
  return conv_launch(step, x, w, bias, y, (float*)ws, B, D, H, W, Cin, Cout, act, 0, (hipStream_t)stream);
}
int modet_conv3d_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                     int D, int H, int W, int Cin, int Cout, int act, modet_stream_t stream, modet_step_ctx_t* step) {
  return conv3d_fwd_impl(x, w, bias, y, ws, ws_bytes, B, D, H, W, Cin, Cout, act, stream, step, true);
}
int modet_conv3d_fwd_bounded(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                             int D, int H, int W, int Cin, int Cout, int act, modet_stream_t stream, modet_step_ctx_t* step) {
  return conv3d_fwd_impl(x, w, bias, y, ws, ws_bytes, B, D, H, W, Cin, Cout, act, stream, step, false);
}
int modet_conv3d_fwd_amax_out(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                              int D, int H, int W, int Cin, int Cout, int act, float* y_amax, modet_stream_t stream,
                              modet_step_ctx_t* step) {
  MODET_CHECK_PTR(y_amax);
  return conv3d_fwd_impl(x, w, bias, y, ws, ws_bytes, B, D, H, W, Cin, Cout, act, stream, step, true, y_amax);
}

// InstanceNorm statistics are fused into the conv epilogue (staged or direct-store) for every Cout the model
// normalises (a multiple of 4, at most 128: 2*Cout columns fit the 256-thread finalize)
static bool conv_stats_ok(int Cin, int Cout) { return Cin != 1 && Cout % 4 == 0 && Cout <= 128; }

// rows per sample of the statistics buffer: the larger persistent grid of the two instantiations (plain / XF)
static int conv_stats_rows(int B, int D, int H, int W, int Cin, int Cout) {
  const int a = conv_grid_x(B, D, H, W, Cin, Cout, false), b = conv_grid_x(B, D, H, W, Cin, Cout, true);
  return a > b ? a : b;
}

size_t modet_conv3d_normin_stats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  if (!conv_stats_ok(Cin, Cout) || B > 32) return 0;
  if (use_x3(B, D, H, W, Cin, Cout)) return modetx_x3_stats_bytes(B, D, H, W, Cin, Cout);
  if (use_q(B, D, H, W, Cin, Cout)) return modetx_q_stats_bytes(B, D, H, W, Cin, Cout);
  return ((size_t)B * Cout + (size_t)B * conv_stats_rows(B, D, H, W, Cin, Cout) * Cout * 2) * sizeof(float);
}

size_t modet_conv3d_stats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  if (!conv_stats_ok(Cin, Cout) || B > 32) return 0;
  if (use_x3(B, D, H, W, Cin, Cout)) return modetx_x3_stats_bytes(B, D, H, W, Cin, Cout);    // one row per workgroup
  if (use_q(B, D, H, W, Cin, Cout)) return modetx_q_stats_bytes(B, D, H, W, Cin, Cout);      // one row per output tile
  if (use_split(Cin, Cout, (int64_t)B * D * H * W)) return modetx_split_stats_bytes(B, D, H, W, Cin, Cout);     // one row per output tile
  // [sample][Cout] shift header, then [sample][workgroup][Cout][2] partial sums of (y - shift), (y - shift)^2; reduced by
  // modet_instnorm_lrelu_fwd_stats / modet_instnorm_stats
  return ((size_t)B * Cout + (size_t)B * conv_stats_rows(B, D, H, W, Cin, Cout) * Cout * 2) * sizeof(float);
}

static int conv3d_fwd_stats_impl(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                                 float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                 modet_stream_t stream, modet_step_ctx_t* step, bool x_free, const float* x_amax = nullptr) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(w); MODET_CHECK_PTR(y); MODET_CHECK_PTR(ws); MODET_CHECK_PTR(stats);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (!conv_stats_ok(Cin, Cout)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < fwd_ws_elems(Cin, Cout) * sizeof(float)) return MODET_ERR_WORKSPACE;
  if (stats_bytes < modet_conv3d_stats_bytes(B, D, H, W, Cin, Cout)) return MODET_ERR_WORKSPACE;
  if (use_x3(B, D, H, W, Cin, Cout)) {
    if (ws_bytes < modetx_x3_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    hipLaunchKernelGGL(conv_shift_kernel, dim3(cdiv(B * Cout, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       (const float*)nullptr, (const float*)nullptr, stats, B, D, H, W, Cin, Cout);
    return modetx_x3_conv(step, x, w, bias, y, ws, stats, nullptr, nullptr, B, D, H, W, Cin, Cout, 0, 0, (hipStream_t)stream, x_amax, x_free);
  }
  if (use_q(B, D, H, W, Cin, Cout)) {
    if (ws_bytes < modetx_q_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    hipLaunchKernelGGL(conv_shift_kernel, dim3(cdiv(B * Cout, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       (const float*)nullptr, (const float*)nullptr, stats, B, D, H, W, Cin, Cout);
    return modetx_q_conv(step, x, w, bias, y, ws, stats, nullptr, nullptr, B, D, H, W, Cin, Cout, 0, (hipStream_t)stream, nullptr, nullptr, nullptr,
                         nullptr, nullptr, x_free);
  }
  if (use_split(Cin, Cout, (int64_t)B * D * H * W)) {
    if (ws_bytes < modetx_split_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    return modetx_split_conv(step, x, w, bias, y, ws, stats, B, D, H, W, Cin, Cout, 0, (hipStream_t)stream);
  }
  return conv_launch(step, x, w, bias, y, (float*)ws, B, D, H, W, Cin, Cout, 0, 0, (hipStream_t)stream, stats, nullptr,
                     ConvIn{nullptr, nullptr, conv_stats_rows(B, D, H, W, Cin, Cout), nullptr});
}
int modet_conv3d_fwd_stats(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                           float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                           modet_stream_t stream, modet_step_ctx_t* step) {
  return conv3d_fwd_stats_impl(x, w, bias, y, ws, ws_bytes, stats, stats_bytes, B, D, H, W, Cin, Cout, stream, step, true);
}
int modet_conv3d_fwd_stats_bounded(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                                   float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                   modet_stream_t stream, modet_step_ctx_t* step) {
  return conv3d_fwd_stats_impl(x, w, bias, y, ws, ws_bytes, stats, stats_bytes, B, D, H, W, Cin, Cout, stream, step, false);
}
int modet_conv3d_fwd_stats_amax(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                                float* stats, size_t stats_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                const float* x_amax, modet_stream_t stream, modet_step_ctx_t* step) {
  return conv3d_fwd_stats_impl(x, w, bias, y, ws, ws_bytes, stats, stats_bytes, B, D, H, W, Cin, Cout, stream, step, true, x_amax);
}

int modet_conv3d_fwd_normin(const float* x_raw, const float* in_mean, const float* in_rstd, const float* w,
                            const float* bias, float* y, void* ws, size_t ws_bytes, float* stats, size_t stats_bytes, int B,
                            int D, int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step) {
  MODET_CHECK_PTR(x_raw); MODET_CHECK_PTR(in_mean); MODET_CHECK_PTR(in_rstd); MODET_CHECK_PTR(w); MODET_CHECK_PTR(y);
  MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (Cin % 4 != 0 || Cin == 1) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < fwd_ws_elems(Cin, Cout) * sizeof(float)) return MODET_ERR_WORKSPACE;
  if (stats) {
    if (!conv_stats_ok(Cin, Cout)) return MODET_ERR_UNSUPPORTED;
    if (stats_bytes < modet_conv3d_normin_stats_bytes(B, D, H, W, Cin, Cout)) return MODET_ERR_WORKSPACE;
  }
  if (use_x3(B, D, H, W, Cin, Cout)) {
    if (ws_bytes < modetx_x3_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    if (stats)
      hipLaunchKernelGGL(conv_shift_kernel, dim3(cdiv(B * Cout, 4)), dim3(256), 0, (hipStream_t)stream, x_raw, w, bias, in_mean,
                         in_rstd, stats, B, D, H, W, Cin, Cout);
    return modetx_x3_conv(step, x_raw, w, bias, y, ws, stats, in_mean, in_rstd, B, D, H, W, Cin, Cout, 0, 0, (hipStream_t)stream);
  }
  if (use_q(B, D, H, W, Cin, Cout)) {
    if (ws_bytes < modetx_q_ws_bytes(Cin, Cout)) return MODET_ERR_WORKSPACE;
    if (stats)
      hipLaunchKernelGGL(conv_shift_kernel, dim3(cdiv(B * Cout, 4)), dim3(256), 0, (hipStream_t)stream, x_raw, w, bias, in_mean,
                         in_rstd, stats, B, D, H, W, Cin, Cout);
    return modetx_q_conv(step, x_raw, w, bias, y, ws, stats, in_mean, in_rstd, B, D, H, W, Cin, Cout, 0, (hipStream_t)stream);
  }
  return conv_launch(step, x_raw, w, bias, y, (float*)ws, B, D, H, W, Cin, Cout, 0, 0, (hipStream_t)stream, stats, nullptr,
                     ConvIn{in_mean, in_rstd, stats ? conv_stats_rows(B, D, H, W, Cin, Cout) : 0, nullptr});
}

size_t modet_conv3d_bwd_data_instats_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  if (B > 32 || Cin % 4 != 0) return 0;
  if (use_x3(B, D, H, W, Cout, Cin)) return modetx_x3_bst_rows_bytes(B, D, H, W, Cin, Cout);
  // the channel-quad kernel (family 5) carries the same epilogue; 2 Cin <= 256: the rows' finalize sums 2 Cin columns
  if (use_q(B, D, H, W, Cout, Cin) && 2 * Cin <= 256) return modetx_q_bst_rows_bytes(B, D, H, W, Cout, Cin);
  return 0;
}

int modet_conv3d_bwd_data_instats(const float* d_y, const float* w, float* d_x, const float* x_raw, const float* mean,
                                  const float* rstd, float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B, int D,
                                  int H, int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step) {
  return modet_conv3d_bwd_data_instats_amax(d_y, w, d_x, x_raw, mean, rstd, rows, rows_bytes, ws, ws_bytes, B, D, H, W, Cin, Cout,
                                            nullptr, stream, step);
}

int modet_conv3d_bwd_data_instats_amax(const float* d_y, const float* w, float* d_x, const float* x_raw, const float* mean,
                                       const float* rstd, float* rows, size_t rows_bytes, void* ws, size_t ws_bytes, int B, int D,
                                       int H, int W, int Cin, int Cout, const float* dy_amax, modet_stream_t stream,
                                       modet_step_ctx_t* step) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(w); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(ws);
  MODET_CHECK_PTR(x_raw); MODET_CHECK_PTR(mean); MODET_CHECK_PTR(rstd); MODET_CHECK_PTR(rows);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  const size_t need = modet_conv3d_bwd_data_instats_bytes(B, D, H, W, Cin, Cout);
  if (need == 0) return MODET_ERR_UNSUPPORTED;
  if (rows_bytes < need) return MODET_ERR_WORKSPACE;
  if (!use_x3(B, D, H, W, Cout, Cin)) {
    if (ws_bytes < modetx_q_ws_bytes(Cout, Cin)) return MODET_ERR_WORKSPACE;
    return modetx_q_conv(step, d_y, w, nullptr, d_x, ws, nullptr, nullptr, nullptr, B, D, H, W, Cout, Cin, 1, (hipStream_t)stream,
                         dy_amax, x_raw, mean, rstd, rows);
  }
  if (ws_bytes < fwd_ws_elems(Cout, Cin) * sizeof(float) || ws_bytes < modetx_x3_ws_bytes(Cout, Cin)) return MODET_ERR_WORKSPACE;
  return modetx_x3_dgrad_bst(step, d_y, w, d_x, x_raw, mean, rstd, rows, ws, B, D, H, W, Cin, Cout, (hipStream_t)stream, dy_amax);
}

int modet_conv3d_bwd_data(const float* d_y, const float* w, float* d_x, void* ws, size_t ws_bytes, int B, int D, int H,
                          int W, int Cin, int Cout, modet_stream_t stream, modet_step_ctx_t* step) {
  return modet_conv3d_bwd_data_amax(d_y, w, d_x, ws, ws_bytes, B, D, H, W, Cin, Cout, nullptr, stream, step);
}

int modet_conv3d_bwd_data_amax(const float* d_y, const float* w, float* d_x, void* ws, size_t ws_bytes, int B, int D, int H,
                               int W, int Cin, int Cout, const float* dy_amax, modet_stream_t stream, modet_step_ctx_t* step) {
  MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(w); MODET_CHECK_PTR(d_x); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (ws_bytes < fwd_ws_elems(Cout, Cin) * sizeof(float)) return MODET_ERR_WORKSPACE;
  // a convolution of d_y (Cout channels) producing Cin channels
  if (use_x3(B, D, H, W, Cout, Cin)) {
    if (ws_bytes < modetx_x3_ws_bytes(Cout, Cin)) return MODET_ERR_WORKSPACE;
    return modetx_x3_conv(step, d_y, w, nullptr, d_x, ws, nullptr, nullptr, nullptr, B, D, H, W, Cout, Cin, 0, 1, (hipStream_t)stream,
                          dy_amax);
  }
  if (use_q(B, D, H, W, Cout, Cin)) {
    if (ws_bytes < modetx_q_ws_bytes(Cout, Cin)) return MODET_ERR_WORKSPACE;
    return modetx_q_conv(step, d_y, w, nullptr, d_x, ws, nullptr, nullptr, nullptr, B, D, H, W, Cout, Cin, 1, (hipStream_t)stream,
                         dy_amax);
  }
  if (use_split(Cout, Cin, (int64_t)B * D * H * W)) {
    if (ws_bytes < modetx_split_ws_bytes(Cout, Cin)) return MODET_ERR_WORKSPACE;
    return modetx_split_conv(step, d_y, w, nullptr, d_x, ws, nullptr, B, D, H, W, Cout, Cin, 1, (hipStream_t)stream);
  }
  return conv_launch(step, d_y, w, nullptr, d_x, (float*)ws, B, D, H, W, Cout, Cin, 0, 1, (hipStream_t)stream);
}

size_t modet_conv3d_bwd_weight_ws_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  const WgPlan p = plan_wgrad(B, D, H, W, Cin, Cout);
  // sized for the largest persistent grid (1024 workgroups) so the value does not depend on the occupancy query
  const int gx = p.gy >= 1024 ? 1 : 1024 / p.gy;
  const size_t fl = (size_t)gx * p.gy * p.ng * 256;
  const size_t c1 = (size_t)C1_MFMA_BLOCKS * 4 * 2 * 256;       // per-wave partial tiles of conv_c1_wgrad_mfma_kernel
  size_t n = (fl > c1 ? fl : c1) * sizeof(float);
  if (use_x3_wgrad(B, D, H, W, Cin, Cout)) {
    const size_t x3 = modetx_x3_wgrad_ws_bytes(B, D, H, W, Cin, Cout);
    n = n > x3 ? n : x3;
  }
  if (use_wtr_wgrad(B, D, H, W, Cin, Cout)) {
    const size_t w3 = modetx_wtr_ws_bytes(B, D, H, W, Cin, Cout);
    n = n > w3 ? n : w3;
  }
  return n;
}

static int conv_bwd_weight_impl(const float* x, const float* d_y, const float* y_act, float* d_w, float* d_bias, void* ws,
                                size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                modet_step_ctx* defer = nullptr, const float* dy_amax = nullptr, const float* x_amax = nullptr);

// ---- deferred reductions: modet_conv3d_bwd_weight*_defer only produce the partial tiles (the workspace must stay
// untouched until the flush) and queue the reduction in the caller's step context; modet_conv3d_wgrad_defer_flush runs
// everything queued there at once.  The context's mutex covers the queue: the autograd engine calls backward from its
// own thread.
static void reduce_or_defer(const ReduceJob& j, int blocks, hipStream_t s, modet_step_ctx* defer) {
  if (defer) {
    std::lock_guard<std::mutex> lk(defer->mu);
    defer->rjobs.push_back(j);
    defer->rblocks.push_back(blocks);
    return;
  }
  if (j.mode == 0)
    hipLaunchKernelGGL(wgrad_reduce_kernel<0>, dim3(blocks), dim3(256), 0, s, j.part, j.dw, j.db, j.Cin, j.Cout, j.gx, j.gy,
                       j.n_ci, j.cit, j.ng);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, j.part, j.dw, j.db, j.Cin, j.Cout, j.gx, j.gy,
                       j.n_ci, j.cit, j.ng);
}

int modet_conv3d_wgrad_defers_operands(int B, int D, int H, int W, int Cin, int Cout) {
  return use_wtr_wgrad(B, D, H, W, Cin, Cout) && modetx_wtr_batches(B, D, H, W) ? 1 : 0;
}

int modet_conv3d_wgrad_defer_flush(modet_step_ctx_t* c, modet_stream_t stream) {
  MODET_CHECK_PTR(c);
  modetx_wtr_flush(c, (hipStream_t)stream);               // queued partial-tile launches, one grid per kernel variant ...
  modetx_bf16_defer_flush(c, (hipStream_t)stream);        // ... then the queue of reductions (modet_conv3d_bf16_bwd_weight_defer's too)
  std::vector<ReduceJob> jobs;
  std::vector<int> blocks;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    jobs.swap(c->rjobs);
    blocks.swap(c->rblocks);
  }
  for (size_t i0 = 0; i0 < jobs.size(); i0 += REDUCE_MAX_JOBS) {
    ReduceTable t;
    const int n = (int)(jobs.size() - i0 < (size_t)REDUCE_MAX_JOBS ? jobs.size() - i0 : (size_t)REDUCE_MAX_JOBS);
    int total = 0;
    for (int i = 0; i < n; ++i) { t.job[i] = jobs[i0 + i]; t.first[i] = total; total += blocks[i0 + i]; }
    for (int i = n; i <= REDUCE_MAX_JOBS; ++i) t.first[i] = total;
    t.n = n;
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, t);
  }
  return modet_launch_status();
}

int modet_conv3d_bwd_weight(const float* x, const float* d_y, float* d_w, float* d_bias, void* ws, size_t ws_bytes,
                            int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream) {
  return conv_bwd_weight_impl(x, d_y, nullptr, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream);
}

int modet_conv3d_bwd_weight_defer(const float* x, const float* d_y, const float* y_act, float* d_w, float* d_bias, void* ws,
                                  size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                  modet_step_ctx_t* step) {
  MODET_CHECK_PTR(step);
  if (y_act && !(Cin == 1 && Cout == 4)) return MODET_ERR_UNSUPPORTED;
  return conv_bwd_weight_impl(x, d_y, y_act, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream, step);
}

int modet_conv3d_bwd_weight_amax(const float* x, const float* d_y, float* d_w, float* d_bias, void* ws, size_t ws_bytes, int B,
                                 int D, int H, int W, int Cin, int Cout, const float* dy_amax, modet_stream_t stream,
                                 modet_step_ctx_t* step) {
  return conv_bwd_weight_impl(x, d_y, nullptr, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream, step, dy_amax);
}

int modet_conv3d_bwd_weight_amax2(const float* x, const float* d_y, float* d_w, float* d_bias, void* ws, size_t ws_bytes, int B,
                                  int D, int H, int W, int Cin, int Cout, const float* dy_amax, const float* x_amax,
                                  modet_stream_t stream, modet_step_ctx_t* step) {
  if (x_amax && !use_x3_wgrad(B, D, H, W, Cin, Cout)) dy_amax = nullptr;      // (only that family scales x by its maximum: else bf16x3)
  return conv_bwd_weight_impl(x, d_y, nullptr, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream, step, dy_amax, x_amax);
}

int modet_conv3d_bwd_weight_normin_ok(int B, int D, int H, int W, int Cin, int Cout) {
  return use_x3_wgrad(B, D, H, W, Cin, Cout) ? 1 : 0;
}

int modet_conv3d_bwd_weight_normin(const float* x_raw, const float* in_mean, const float* in_rstd, const float* d_y, float* d_w,
                                   float* d_bias, void* ws, size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                   const float* dy_amax, modet_stream_t stream, modet_step_ctx_t* step) {
  MODET_CHECK_PTR(x_raw); MODET_CHECK_PTR(in_mean); MODET_CHECK_PTR(in_rstd); MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(d_w); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (!use_x3_wgrad(B, D, H, W, Cin, Cout)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_conv3d_bwd_weight_ws_bytes(B, D, H, W, Cin, Cout)) return MODET_ERR_WORKSPACE;
  return modetx_x3_wgrad(step, x_raw, d_y, d_w, d_bias, ws, B, D, H, W, Cin, Cout, (hipStream_t)stream, dy_amax, in_mean, in_rstd);
}

int modet_conv3d_bwd_weight_act(const float* x, const float* d_y, const float* y_act, float* d_w, float* d_bias,
                                void* ws, size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout,
                                modet_stream_t stream) {
  MODET_CHECK_PTR(y_act);
  if (!(Cin == 1 && Cout == 4)) return MODET_ERR_UNSUPPORTED;     // only the first encoder block (ConvBlock 1 -> 4)
  return conv_bwd_weight_impl(x, d_y, y_act, d_w, d_bias, ws, ws_bytes, B, D, H, W, Cin, Cout, stream);
}

static int conv_bwd_weight_impl(const float* x, const float* d_y, const float* y_act, float* d_w, float* d_bias, void* ws,
                                size_t ws_bytes, int B, int D, int H, int W, int Cin, int Cout, modet_stream_t stream,
                                modet_step_ctx* defer, const float* dy_amax, const float* x_amax) {
  MODET_CHECK_PTR(x); MODET_CHECK_PTR(d_y); MODET_CHECK_PTR(d_w); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  if (Cout > NTHR) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < modet_conv3d_bwd_weight_ws_bytes(B, D, H, W, Cin, Cout)) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1 && Cout == 4) {
    const int tx = cdiv(W, TX), ty = cdiv(H, WG_TY), tz = cdiv(D, C1_TZ);
    const int ntiles = B * tx * ty * tz;
    const int nblk = ntiles < C1_MFMA_BLOCKS ? ntiles : C1_MFMA_BLOCKS;
    hipLaunchKernelGGL(conv_c1_wgrad_mfma_kernel, dim3(nblk), dim3(NTHR), 0, s, x, d_y, y_act, (float*)ws, D, H, W, tx, ty, tz,
                       ntiles);
    reduce_or_defer(ReduceJob{(const float*)ws, d_w, d_bias, 1, 4, nblk, 1, 1, 1, 2, 0}, 2 * 4, s, defer);
    return modet_launch_status();
  }
  if (!y_act && use_x3_wgrad(B, D, H, W, Cin, Cout))
    return modetx_x3_wgrad(defer, x, d_y, d_w, d_bias, ws, B, D, H, W, Cin, Cout, s, dy_amax, nullptr, nullptr, x_amax);
  if (!y_act && use_wtr_wgrad(B, D, H, W, Cin, Cout))
    return modetx_wtr_wgrad(defer, x, d_y, d_w, d_bias, ws, B, D, H, W, Cin, Cout, s, dy_amax);
  const WgPlan p = plan_wgrad(B, D, H, W, Cin, Cout);
  float* part = (float*)ws;
  if (p.np) {
    const bool rowld = Cin == 8 && Cout == 8;
#define NP_LAUNCH(TZ_, R_) hipLaunchKernelGGL((conv3d_wgrad_np_kernel<8, TZ_, R_>), dim3(p.gx), dim3(NTHR), 0, s, x, d_y, part, D, \
                                               H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, p.ntiles)
    if (p.tz == 4) { if (rowld) NP_LAUNCH(4, true); else NP_LAUNCH(4, false); }
    else { if (rowld) NP_LAUNCH(2, true); else NP_LAUNCH(2, false); }
#undef NP_LAUNCH
    reduce_or_defer(ReduceJob{(const float*)part, d_w, d_bias, Cin, Cout, p.gx, 1, 1, p.cit, p.ng, 1}, p.ng * 4, s, defer);
    return modet_launch_status();
  }
  dim3 grid(p.gx, p.gy);
#define WG_LAUNCH(CIT_, TZ_) hipLaunchKernelGGL((conv3d_wgrad_kernel<CIT_, TZ_>), grid, dim3(NTHR), 0, s, x, d_y, part, D, H, W, \
                                               Cin, Cout, p.tiles_x, p.tiles_y, p.tiles_z, p.ntiles, p.n_ci)
  if (p.cit == 4) { if (p.tz == 4) WG_LAUNCH(4, 4); else WG_LAUNCH(4, 2); }
  else if (p.cit == 8) { if (p.tz == 4) WG_LAUNCH(8, 4); else WG_LAUNCH(8, 2); }
  else WG_LAUNCH(16, 2);
#undef WG_LAUNCH
  reduce_or_defer(ReduceJob{(const float*)part, d_w, d_bias, Cin, Cout, p.gx, p.gy, p.n_ci, p.cit, p.ng, 0}, p.gy * p.ng * 4, s, defer);
  return modet_launch_status();
}

}  // extern "C"
