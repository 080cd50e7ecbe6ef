// micro-benchmark: what does a global float atomic cost on gfx950 -- per lane, per 32-byte segment, per instruction, and
// when a partial-line atomic follows a full-line one?   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/l2_atomic.hip -o /tmp/l2_atomic && /tmp/l2_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// buffer of N floats viewed as rows of 64 floats (256 B).  One wave handles `rows_per_wave` consecutive rows.
// MODE 0: dense      -- every lane adds to its own float of the row (one 256 B instruction per row)
// MODE 1: sparse8    -- only lanes 0..7 add (32 B per instruction), same number of instructions as MODE 0
// MODE 2: dense + sparse8 into the NEXT row's first 32 B right after (the "leftover meets the neighbour's line" pattern)
// MODE 3: dense + sparse8 into the SAME row's last 32 B (same line, same wave)
// MODE 4: two dense adds to the same row back to back
// MODE 5: strided32  -- lane group g (8 lanes) adds to row*64 + perm(g)*... 8 segments of 32 B in 8 DIFFERENT rows (scattered)
// MODE 6: plain stores dense (reference)
// MODE 7: sparse8 only but at stride so that every instruction hits a different 128 B line (no neighbour reuse)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, int64_t nrows, int rows_per_wave, float v) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t r0 = wave * rows_per_wave;
  for (int i = 0; i < rows_per_wave; ++i) {
    const int64_t r = r0 + i;
    if (r >= nrows) return;
    float* p = buf + r * 64;
    if (MODE == 0) atomicAdd(p + lane, v);
    if (MODE == 1) { if (lane < 8) atomicAdd(p + lane, v); }
    if (MODE == 2) { atomicAdd(p + lane, v); if (lane < 8 && r + 1 < nrows) atomicAdd(p + 64 + lane, v); }
    if (MODE == 3) { atomicAdd(p + lane, v); if (lane >= 56) atomicAdd(p + lane, v); }
    if (MODE == 4) { atomicAdd(p + lane, v); atomicAdd(p + lane, v); }
    if (MODE == 5) { const int g = lane >> 3; const int64_t rr = (r + (int64_t)g * 4099) % nrows; atomicAdd(buf + rr * 64 + g * 8 + (lane & 7), v); }
    if (MODE == 6) p[lane] = v;
    if (MODE == 7) { if (lane < 8) atomicAdd(p + lane + ((r & 1) ? 32 : 0), v); }
  }
}
template <int MODE> float run(float* buf, int64_t nrows, int rpw) {
  const int64_t waves = (nrows + rpw - 1) / rpw;
  const int blocks = (int)((waves + 3) / 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, nrows, rpw, 1.0f);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, nrows, rpw, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  const int64_t n = 39321600;            // 160*192*160*8 floats = 157 MB
  const int64_t nrows = n / 64;
  float* buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
  for (int rpw : {1, 8}) {
    printf("rows per wave %d (%.1f M rows of 256 B)\n", rpw, nrows / 1e6);
    printf("  0 dense 256B/instr            %.3f ms\n", run<0>(buf, nrows, rpw));
    printf("  1 sparse 32B/instr            %.3f ms\n", run<1>(buf, nrows, rpw));
    printf("  2 dense + 32B into next row   %.3f ms\n", run<2>(buf, nrows, rpw));
    printf("  3 dense + 32B same row        %.3f ms\n", run<3>(buf, nrows, rpw));
    printf("  4 dense x2 same row           %.3f ms\n", run<4>(buf, nrows, rpw));
    printf("  5 8 x 32B scattered rows      %.3f ms\n", run<5>(buf, nrows, rpw));
    printf("  6 plain dense stores          %.3f ms\n", run<6>(buf, nrows, rpw));
    printf("  7 sparse 32B, alternating     %.3f ms\n", run<7>(buf, nrows, rpw));
  }
  return 0;
}
