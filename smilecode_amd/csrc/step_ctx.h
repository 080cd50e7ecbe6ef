// Caller-owned step context of the step-batching entry points (include/modet_hip.h: modet_step_ctx_*,
// modet_conv3d_prepack_*, modet_conv3d_bwd_weight_defer / modet_conv3d_wgrad_defer_flush).  Internal, not part of the ABI:
// the ABI only sees the opaque handle.  Host memory only -- job descriptions (pointers + geometry), never device memory.
// One context = the recorded weight-packing jobs of ONE forward+backward pass and the weight-gradient reductions queued
// by the backward pass in flight.  Nothing here is process-wide: two trainers (two threads, two devices, a training and
// an EMA model) each pass their own context and never see each other's jobs.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <vector>

// fp32 exact-MFMA convs (conv3d.hip): operand layout [tapP][CinP][CoutP]
struct PackKey {
  const float* w; int Cin, Cout, CinP, CoutP, mode, P;
  bool operator==(const PackKey& o) const {
    return w == o.w && Cin == o.Cin && Cout == o.Cout && CinP == o.CinP && CoutP == o.CoutP && mode == o.mode && P == o.P;
  }
};
// bf16 / split convs (conv3d_bf16.hip, conv3d_x3.hip): operand layout [stage][k-step][CoutP][32] x npiece; `layout`
// distinguishes the kernels' packings (0: conv3d_bf16 k = (tap, cin); 1: conv3d_x3 row-packed)
struct PackBKey {
  const float* w; int Cin, Cout, CoutP, CK, nstage, ksteps, mode, npiece, layout;
  bool operator==(const PackBKey& o) const {
    return w == o.w && Cin == o.Cin && Cout == o.Cout && CoutP == o.CoutP && CK == o.CK && nstage == o.nstage &&
           ksteps == o.ksteps && mode == o.mode && npiece == o.npiece && layout == o.layout;
  }
};
// queued reductions of weight-gradient partial tiles
struct ReduceJob {
  const float* part; float* dw; float* db;
  int Cin, Cout, gx, gy, n_ci, cit, ng, mode;
};
struct BRedJob {
  const float* part; float* red; float* dw; float* dbias;
  int64_t row_fl;
  int gx, Cin, Cout, cib, u, ntb, gy, n_coblk;
  int layout;   // 0: slot = group * 3 + dx, n = cout (conv3d_bf16_wgrad_kernel, conv_x3_wgrad_kernel<.., NP = false>)
                // 1: slot = group * 2 + t, n = q * 8 + cout (conv_x3_wgrad_kernel<.., NP = true>: d_y shifted by q voxels)
                // 2: conv_wgrad_tr_kernel (conv3d_wtr.hip): cib = NQ, u = MT, ntb = NT; M tile = 4 chunks of q = tap * NQ + quad
};

// a queued conv_wgrad_tr_kernel launch (conv3d_wtr.hip): the flush groups them by kernel variant (nq, nt, vec) into one grid each
struct WtrQueued {
  const float* x; const float* dy; float* part;
  int B, D, H, W, Cin, Cout, tiles_x, tiles_y, tiles_z, ntiles, n_coblk;
  int nq, nt, vec, gx, gy;
  const float* amax;                  // two f16 pieces: the maxima of |d_y| (MODET_AMAX_SLOTS slots), else null (three bf16 pieces)
};

struct modet_step_ctx {
  std::mutex mu;                      // forward runs on the caller's thread, backward on the autograd engine's
  bool recording = false, active = false;
  std::vector<PackKey> jobs;          // recorded fp32 packing jobs, in launch order
  std::vector<size_t> off;            // float offset of each job's packed weights inside the arena
  float* arena = nullptr;
  std::vector<PackBKey> bjobs;        // recorded 16-bit packing jobs (they follow the fp32 ones in the arena)
  std::vector<size_t> boff;           // ushort offset inside the 16-bit part of the arena
  unsigned short* barena = nullptr;
  std::vector<ReduceJob> rjobs;       // deferred fp32 weight-gradient reductions
  std::vector<int> rblocks;
  std::vector<BRedJob> brjobs;        // deferred bf16 weight-gradient reductions
  std::vector<WtrQueued> wtrjobs;     // deferred partial-tile launches of the many-channel weight gradients
};
