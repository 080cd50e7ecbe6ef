// d_src of the trilinear warp's backward WITHOUT global float atomics (SpatialTransformer backward, reference
// ModeT/models.py:55-67 -> ATen grid_sampler_3d_backward scatters with atomicAdd): round 5's prototype
// (tools/micro/warp_tile_proto.hip, profiles/r05y_*, r05z_*) as an entry point.  OFF by default in the Python layer
// (ops.WARP_TILE_DSRC): it has its op-level parity tests, not yet a full-suite pass with the step routed through it.
//   Why it exists: the shipped scatter is bound by the L2 float-atomic unit (18 G sector-atomics/s on the model's flow), and the
//   LDS privatisations of rounds 1 / 4 lost because ds_add_f32 retires 0.33 lanes/clk/CU -- the INTEGER LDS atomics run 20-37 x
//   faster (ds_add_u64 6.9-12.3 lanes/clk/CU, profiles/r05y_lds_atomic_microbench.txt).
//   A  bin the source voxels by the 8^3 DESTINATION tile of their base corner: per source workgroup an LDS hash histogram
//      (integer LDS atomics), one returning global atomic per (workgroup, tile); twice: count -> exclusive scan -> fill
//   B  one workgroup per destination tile: its list -> a 9^3 x 8-channel window of 64-bit FIXED-POINT sums in LDS (scale from
//      max |d_out|: 2^-30 of it per contribution, finer than fp32), laid out [channel][cell] (as [cell][channel] a wave's lanes
//      pile onto two banks: 0.33 instead of 0.09 ms of atomics at level 1); the 8^3 owned cells leave as plain stores (no zero
//      fill of d_src), the 217 high-face cells go to a side buffer; C > 8 in channel passes through the same window
//   C  every owned cell on a low tile face adds the neighbours' side-buffer cells (gather)
// Integer sums: the result does not depend on the order of the lists -> bit-reproducible.  Every launch is a kernel with fixed
// arguments (no memset node, no host read of max |d_out|): capturable into a hipGraph.
#include "common.h"

namespace {
constexpr int TL = 8, WN = 9, CELLS = WN * WN * WN, NBORDER = CELLS - TL * TL * TL;      // 729 window cells, 217 on the high faces
constexpr int SZ = 4, SY = 8, SX = 32, SVOX = SZ * SY * SX;                             // source block of pass A: 1024 voxels
constexpr int HASH = 256;
struct Geo { int D, H, W, C, tz, ty, tx, ntiles, B; };      // ntiles = tiles per sample

struct Entry { int tile, bz, by, bx; float fz, fy, fx; };
__device__ __forceinline__ bool make_entry(const float* __restrict__ flow, int64_t p, int z, int y, int x, const Geo g, Entry& e) {
  const float pz = (float)z + flow[p * 3], py = (float)y + flow[p * 3 + 1], px = (float)x + flow[p * 3 + 2];
  const float flz = floorf(pz), fly = floorf(py), flx = floorf(px);
  if (!(flz >= -1.f && flz <= (float)(g.D - 1) && fly >= -1.f && fly <= (float)(g.H - 1) && flx >= -1.f && flx <= (float)(g.W - 1)))
    return false;                                   // every corner outside the volume (or a non-finite flow)
  e.bz = (int)flz; e.by = (int)fly; e.bx = (int)flx;
  e.fz = pz - flz; e.fy = py - fly; e.fx = px - flx;
  const int cz = (e.bz < 0 ? 0 : e.bz) >> 3, cy = (e.by < 0 ? 0 : e.by) >> 3, cx = (e.bx < 0 ? 0 : e.bx) >> 3;
  e.tile = (cz * g.ty + cy) * g.tx + cx;
  return true;
}

// ---- A: FILL = false: tile_count[tile] += entries;  FILL = true: cursor[tile] (initialised to the tile's list offset) hands out
// a segment per (workgroup, tile) and the voxel indices are written there
template <bool FILL>
__global__ __launch_bounds__(256) void bin_kernel(const float* __restrict__ flow, unsigned* __restrict__ counter, int* __restrict__ list,
                                                  const Geo g, int bx_n, int by_n, float* __restrict__ dflow = nullptr,
                                                  const float* __restrict__ dflow_add = nullptr) {
  __shared__ int keys[HASH];
  __shared__ unsigned cnt[HASH], off[HASH];
  const int tid = threadIdx.x;
  keys[tid] = -1; cnt[tid] = 0;
  __syncthreads();
  int t = blockIdx.x;
  const int x0 = (t % bx_n) * SX; t /= bx_n;
  const int y0 = (t % by_n) * SY;
  const int z0 = (t / by_n) * SZ;
  const int x = x0 + (tid & 31), y = y0 + (tid >> 5);
  const int b = blockIdx.y;                                   // sample: its tiles are [b * ntiles, (b + 1) * ntiles)
  flow += (int64_t)b * g.D * g.H * g.W * 3;
  int slot[SZ];
  unsigned rank[SZ];
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    const int z = z0 + k;
    slot[k] = -1; rank[k] = 0;
    if (x >= g.W || y >= g.H || z >= g.D) continue;
    const int64_t p = ((int64_t)z * g.H + y) * g.W + x;
    Entry e;
    if (!make_entry(flow, p, z, y, x, g, e)) {
      // no corner inside the volume: no entry; the fused form's d_flow of this voxel is just the second gradient (or 0)
      if (FILL && dflow) {
        const int64_t o = ((int64_t)b * g.D * g.H * g.W + p) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) dflow[o + a] = dflow_add ? dflow_add[o + a] : 0.f;
      }
      continue;
    }
    e.tile += b * g.ntiles;
    int s = (e.tile * 40503) & (HASH - 1);
    for (int probe = 0; probe < HASH; ++probe) {
      const int k0 = atomicCAS(&keys[s], -1, e.tile);
      if (k0 == -1 || k0 == e.tile) break;
      s = (s + 1) & (HASH - 1);
    }
    slot[k] = s;
    rank[k] = atomicAdd(&cnt[s], 1u);
  }
  __syncthreads();
  if (keys[tid] >= 0) off[tid] = atomicAdd(&counter[keys[tid]], cnt[tid]);
  if (!FILL) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SZ; ++k) {
    if (slot[k] < 0) continue;
    list[off[slot[k]] + rank[k]] = ((z0 + k) << 20) | (y << 10) | x;          // (dimensions <= 1024)
  }
}

// exclusive scan of the tile counts (one workgroup): offsets[t], cursor[t] = offsets[t]
__global__ __launch_bounds__(1024) void scan_kernel(const unsigned* __restrict__ count, unsigned* __restrict__ offsets,
                                                    unsigned* __restrict__ cursor, int n) {
  __shared__ unsigned part[1024];
  const int per = (n + 1023) / 1024, b = threadIdx.x * per;
  unsigned s = 0;
  for (int i = 0; i < per; ++i) if (b + i < n) s += count[b + i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int i = 0; i < per; ++i)
    if (b + i < n) { offsets[b + i] = run; cursor[b + i] = run; run += count[b + i]; }
}

// amax[0] = bits of max |x| (non-negative floats order like their bit patterns); amax[0] must be zero on entry
__global__ __launch_bounds__(256) void tile_absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ amax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i * 4 < n; i += (int64_t)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {                                     // ONE atomic per workgroup (8 192 same-address atomics cost 0.1 ms)
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (m > 0.f) atomicMax(amax, __float_as_uint(m));
  }
}
// scale = 2^(30 - e) with max |d_out| = m 2^e, m in [0.5, 1): |d_out| * scale < 2^30 (an int32 per contribution, int64 sums)
__device__ __forceinline__ void fx_scales(const unsigned* amax, float& scale, float& inv) {
  const float a = __uint_as_float(amax[0]);
  int e = 0;
  (void)frexpf(a > 0.f && a < 3.0e38f ? a : 1.f, &e);
  scale = ldexpf(1.f, 30 - e); inv = ldexpf(1.f, e - 30);
}

__device__ __forceinline__ int border_index(int lz, int ly, int lx) {      // cells with max(l) == 8
  if (lz == 8) return ly * 9 + lx;                   // 81
  if (ly == 8) return 81 + lz * 9 + lx;              // 72 (lz < 8)
  return 153 + lz * 8 + ly;                          // 64 (lx == 8, lz, ly < 8)
}

// ---- B: the tile's list -> fixed-point window -> owned cells (plain stores) + high-face cells (side buffer).  ONE lane per
// entry, all eight channels (two float4 of d_out; C == 8): the corner / weight arithmetic of an entry is done once; an entry
// whose d_out is all zero (the step's d_out is: background) costs its loads only; U entries per lane in flight.
__device__ __forceinline__ float fx_to_float(unsigned long long u, float inv_scale) {
  const long long x = (long long)u;
  const unsigned long long a = x < 0 ? (unsigned long long)(-x) : (unsigned long long)x;      // sign-magnitude: no cancellation
  const float m = fmaf((float)(unsigned)(a >> 32), 4294967296.f, (float)(unsigned)a);
  return (x < 0 ? -m : m) * inv_scale;
}
// FLOW (C == 8 only): d_flow as well, from the SAME window -- the eight corners of an entry are cells of its tile's window, so the
// tile's src cells are staged in LDS once (coalesced, every src value read from HBM once) and d_flow[p] = sum over the corners of
// (+-) the other two weights x <src[corner], d_out[p]> needs no gather from memory; + dflow_add[p] (a second gradient of the flow).
template <bool FLOW>
__global__ __launch_bounds__(256) void accumulate_kernel(const float* __restrict__ flow, const float* __restrict__ dout,
                                                         const unsigned* __restrict__ offsets, const unsigned* __restrict__ count,
                                                         const int* __restrict__ list, float* __restrict__ dsrc,
                                                         float* __restrict__ border, const Geo g, const unsigned* __restrict__ amax,
                                                         const float* __restrict__ src = nullptr, float* __restrict__ dflow = nullptr,
                                                         const float* __restrict__ dflow_add = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned long long win[CELLS * 8];
  __shared__ float swin[FLOW ? CELLS * 8 : 1];
  const int tid = threadIdx.x;
  float scale, inv_scale;
  fx_scales(amax, scale, inv_scale);
  const int tile = blockIdx.x;                                // over all samples
  const int b = tile / g.ntiles;
  int t = tile - b * g.ntiles;
  const int ox = (t % g.tx) * TL; t /= g.tx;
  const int oy = (t % g.ty) * TL;
  const int oz = (t / g.ty) * TL;
  const unsigned n = count[tile], base = offsets[tile];
  const int64_t V = (int64_t)g.D * g.H * g.W;
  flow += (int64_t)b * V * 3;
  dout += (int64_t)b * V * g.C;
  dsrc += (int64_t)b * V * g.C;
  const int C = g.C;
  if constexpr (FLOW) {
    src += (int64_t)b * V * 8; dflow += (int64_t)b * V * 3;
    if (dflow_add) dflow_add += (int64_t)b * V * 3;
    for (int cell = tid; cell < CELLS; cell += 256) {
      const int lx = cell % WN, lr = cell / WN, ly = lr % WN, lz = lr / WN;
      const int gz = oz + lz, gy = oy + ly, gx = ox + lx;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (gz < g.D && gy < g.H && gx < g.W) {
        const float* sp = src + (((int64_t)gz * g.H + gy) * g.W + gx) * 8;
        a = *reinterpret_cast<const float4*>(sp); c = *reinterpret_cast<const float4*>(sp + 4);
      }
      swin[0 * CELLS + cell] = a.x; swin[1 * CELLS + cell] = a.y; swin[2 * CELLS + cell] = a.z; swin[3 * CELLS + cell] = a.w;
      swin[4 * CELLS + cell] = c.x; swin[5 * CELLS + cell] = c.y; swin[6 * CELLS + cell] = c.z; swin[7 * CELLS + cell] = c.w;
    }
  }
  // C channels in passes of eight: the window is 46 KB whatever C is (three workgroups per CU); the list and the flow are re-read
  // per pass (L2), d_out once in total
  for (int c0 = 0; c0 < C; c0 += 8) {
  if (c0) __syncthreads();                                    // the previous pass's flush has read the window
  {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2* w2 = reinterpret_cast<u64x2*>(win);
    for (int i = tid; i < CELLS * 4; i += 256) w2[i] = (u64x2){0ull, 0ull};
  }
  __syncthreads();
  constexpr int U = 4;
  for (unsigned i0 = tid; i0 < n; i0 += 256 * U) {
    int pk[U];
    float f0[U], f1[U], f2[U];
    float4 ga[U], gb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const unsigned i = i0 + 256 * u; pk[u] = i < n ? list[base + i] : -1; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = pk[u] < 0 ? 0 : pk[u];
      const int64_t v = ((int64_t)(q >> 20) * g.H + ((q >> 10) & 1023)) * g.W + (q & 1023);
      f0[u] = flow[v * 3]; f1[u] = flow[v * 3 + 1]; f2[u] = flow[v * 3 + 2];
      ga[u] = *reinterpret_cast<const float4*>(dout + v * C + c0);
      gb[u] = *reinterpret_cast<const float4*>(dout + v * C + c0 + 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (pk[u] < 0) continue;
      const float gvv[8] = {ga[u].x, ga[u].y, ga[u].z, ga[u].w, gb[u].x, gb[u].y, gb[u].z, gb[u].w};
      bool any = false;
#pragma unroll
      for (int c = 0; c < 8; ++c) any = any || gvv[c] != 0.f;
      const int64_t vox = ((int64_t)(pk[u] >> 20) * g.H + ((pk[u] >> 10) & 1023)) * g.W + (pk[u] & 1023);
      if (!any) {
        if constexpr (FLOW) {
#pragma unroll
          for (int a = 0; a < 3; ++a) dflow[vox * 3 + a] = dflow_add ? dflow_add[vox * 3 + a] : 0.f;
        }
        continue;
      }
      float gfz = 0.f, gfy = 0.f, gfx = 0.f;
      const float pz = (float)(pk[u] >> 20) + f0[u], py = (float)((pk[u] >> 10) & 1023) + f1[u], px = (float)(pk[u] & 1023) + f2[u];
      const float flz = floorf(pz), fly = floorf(py), flx = floorf(px);
      const int bz = (int)flz, by = (int)fly, bx = (int)flx;
      const float fz = pz - flz, fy = py - fly, fx = px - flx;
      float vs[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) vs[c] = gvv[c] * scale;
      const int cell0 = ((bz - oz) * WN + (by - oy)) * WN + (bx - ox);
      // every corner inside the volume (all but the entries at the volume's faces): no per-corner tests
      const bool inner = bz >= 0 && bz + 1 < g.D && by >= 0 && by + 1 < g.H && bx >= 0 && bx + 1 < g.W;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int dz = c >> 2, dy = (c >> 1) & 1, dx = c & 1;
        if (!inner) {
          const int gz = bz + dz, gy = by + dy, gx = bx + dx;
          if (gz < 0 || gz >= g.D || gy < 0 || gy >= g.H || gx < 0 || gx >= g.W) continue;
        }
        const float w = (dz ? fz : 1.f - fz) * (dy ? fy : 1.f - fy) * (dx ? fx : 1.f - fx);
        unsigned long long* wp = win + (cell0 + (dz * WN + dy) * WN + dx);       // window = [channel][cell]: a wave's lanes (different entries) spread over the banks
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(wp + k * CELLS, (unsigned long long)(long long)__float2int_rn(vs[k] * w));
        if constexpr (FLOW) {
          const float* sp = swin + (cell0 + (dz * WN + dy) * WN + dx);
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) dot = fmaf(sp[k * CELLS], gvv[k], dot);
          const float wz = dz ? fz : 1.f - fz, wy = dy ? fy : 1.f - fy, wx = dx ? fx : 1.f - fx;
          gfz += (dz ? 1.f : -1.f) * wy * wx * dot;
          gfy += (dy ? 1.f : -1.f) * wz * wx * dot;
          gfx += (dx ? 1.f : -1.f) * wz * wy * dot;
        }
      }
      if constexpr (FLOW) {
        if (dflow_add) { gfz += dflow_add[vox * 3]; gfy += dflow_add[vox * 3 + 1]; gfx += dflow_add[vox * 3 + 2]; }
        dflow[vox * 3] = gfz; dflow[vox * 3 + 1] = gfy; dflow[vox * 3 + 2] = gfx;
      }
    }
  }
  __syncthreads();
  // flush: one cell (8 channels = 32 bytes of output) per thread and trip
  for (int cell = tid; cell < CELLS; cell += 256) {
    const int lx = cell % WN, lr = cell / WN, ly = lr % WN, lz = lr / WN;
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = fx_to_float(win[k * CELLS + cell], inv_scale);
    float* dst;
    if (lz < TL && ly < TL && lx < TL) {
      const int gz = oz + lz, gy = oy + ly, gx = ox + lx;
      if (gz >= g.D || gy >= g.H || gx >= g.W) continue;
      dst = dsrc + (((int64_t)gz * g.H + gy) * g.W + gx) * C + c0;
    } else {
      dst = border + ((int64_t)tile * NBORDER + border_index(lz, ly, lx)) * C + c0;
    }
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
  }       // channel pass
}

// ---- C: owned cells on a low face of their tile (169 of 512) += the high-face cells of the up to seven neighbours; every
// side-buffer cell is read exactly once
__global__ __launch_bounds__(256) void border_kernel(float* __restrict__ dsrc, const float* __restrict__ border, const Geo g) {
  const int tile = blockIdx.x;
  const int b = tile / g.ntiles;
  int t = tile - b * g.ntiles;
  const int tx = t % g.tx; t /= g.tx;
  const int ty = t % g.ty, tz = t / g.ty;
  const int C = g.C, q4 = C / 4;                              // a thread = four channels of one face cell
  dsrc += (int64_t)b * g.D * g.H * g.W * C;
  for (int j = threadIdx.x; j < 169 * q4; j += 256) {
    const int ch = (j % q4) * 4, k = j / q4;
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
      if ((dz && (lz || tz == 0)) || (dy && (ly || ty == 0)) || (dx && (lx || tx == 0))) continue;
      const int nt = b * g.ntiles + ((tz - dz) * g.ty + (ty - dy)) * g.tx + (tx - dx);
      const float4 v = *reinterpret_cast<const float4*>(border + ((int64_t)nt * NBORDER + border_index(dz ? 8 : lz, dy ? 8 : ly, dx ? 8 : lx)) * C + ch);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      any = true;
    }
    if (!any) continue;
    float4* dp = reinterpret_cast<float4*>(dsrc + (((int64_t)gz * g.H + gy) * g.W + gx) * C + ch);
    float4 d = *dp;
    d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
    *dp = d;
  }
}
}  // namespace

extern "C" {

// [amax 1 + pad 63][count nt][offsets nt][cursor nt] unsigned, [list B*D*H*W] int, [side buffer nt*217*C] float  (nt = tiles of all samples)
size_t modet_warp_bwd_dsrc_tiles_ws_bytes(int B, int D, int H, int W, int C) {
  if (B < 1 || D < 1 || H < 1 || W < 1 || C < 8 || C % 8 != 0 || D > 1024 || H > 1024 || W > 1024) return 0;
  const size_t nt = (size_t)B * cdiv(D, 8) * cdiv(H, 8) * cdiv(W, 8);
  if ((int64_t)B * D * H * W >= (1ll << 31) || nt >= (1u << 30)) return 0;
  return (64 + 3 * nt) * 4 + (size_t)B * D * H * W * 4 + nt * NBORDER * C * 4 + 256;
}

static int tiles_launch(const float* src, const float* flow, const float* d_out, float* d_src, float* d_flow, const float* d_flow_add,
                        void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(flow); MODET_CHECK_PTR(d_out); MODET_CHECK_PTR(d_src); MODET_CHECK_PTR(ws);
  MODET_CHECK_DIM(B > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  const size_t need = modet_warp_bwd_dsrc_tiles_ws_bytes(B, D, H, W, C);
  if (need == 0 || (d_flow && C != 8)) return MODET_ERR_UNSUPPORTED;
  if (ws_bytes < need) return MODET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  Geo g{D, H, W, C, cdiv(D, 8), cdiv(H, 8), cdiv(W, 8), 0, B};
  g.ntiles = g.tz * g.ty * g.tx;
  const int nt = B * g.ntiles;
  unsigned* amax = (unsigned*)ws;
  unsigned* count = amax + 64;
  unsigned* offsets = count + nt;
  unsigned* cursor = offsets + nt;
  int* list = (int*)(cursor + nt);
  float* border = (float*)(list + (size_t)B * D * H * W);
  const int bx_n = cdiv(W, SX), by_n = cdiv(H, SY), bz_n = cdiv(D, SZ);
  modet_zero_async(amax, (size_t)(64 + nt) * 4, s);
  hipLaunchKernelGGL(bin_kernel<false>, dim3(bx_n * by_n * bz_n, B), dim3(256), 0, s, flow, count, list, g, bx_n, by_n, (float*)nullptr,
                     (const float*)nullptr);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, (const unsigned*)count, offsets, cursor, nt);
  hipLaunchKernelGGL(bin_kernel<true>, dim3(bx_n * by_n * bz_n, B), dim3(256), 0, s, flow, cursor, list, g, bx_n, by_n, d_flow, d_flow_add);
  hipLaunchKernelGGL(tile_absmax_kernel, dim3(1024), dim3(256), 0, s, d_out, (int64_t)B * D * H * W * C, amax);
  if (d_flow)
    hipLaunchKernelGGL(accumulate_kernel<true>, dim3(nt), dim3(256), 0, s, flow, d_out, (const unsigned*)offsets, (const unsigned*)count,
                       (const int*)list, d_src, border, g, (const unsigned*)amax, src, d_flow, d_flow_add);
  else
    hipLaunchKernelGGL(accumulate_kernel<false>, dim3(nt), dim3(256), 0, s, flow, d_out, (const unsigned*)offsets, (const unsigned*)count,
                       (const int*)list, d_src, border, g, (const unsigned*)amax, (const float*)nullptr, (float*)nullptr,
                       (const float*)nullptr);
  hipLaunchKernelGGL(border_kernel, dim3(nt), dim3(256), 0, s, d_src, (const float*)border, g);
  return modet_launch_status();
}

int modet_warp_bwd_dsrc_tiles(const float* flow, const float* d_out, float* d_src, void* ws, size_t ws_bytes, int B, int D, int H,
                              int W, int C, modet_stream_t stream) {
  return tiles_launch(nullptr, flow, d_out, d_src, nullptr, nullptr, ws, ws_bytes, B, D, H, W, C, stream);
}

int modet_warp_bwd_tiles(const float* src, const float* flow, const float* d_out, float* d_src, float* d_flow, const float* d_flow_add,
                         void* ws, size_t ws_bytes, int B, int D, int H, int W, int C, modet_stream_t stream) {
  MODET_CHECK_PTR(src); MODET_CHECK_PTR(d_flow);
  return tiles_launch(src, flow, d_out, d_src, d_flow, d_flow_add, ws, ws_bytes, B, D, H, W, C, stream);
}

}  // extern "C"
