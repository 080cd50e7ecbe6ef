// micro-benchmark: LDS float atomic-add throughput on gfx950 (conflict-free, 2-way and same-address patterns) vs plain
// LDS read-modify-write.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomic.hip -o /tmp/lds_atomic && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  __shared__ float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
  __syncthreads();
  const int t = threadIdx.x;
  // MODE 0: ds_add_f32 conflict-free (lane -> consecutive dwords); 1: stride 2 dwords (2-way); 2: 8 lanes share an address;
  // 3: plain read-modify-write (ds_read + v_add + ds_write), conflict-free
  // round 5: 4: ds_add_u32 conflict-free (integer, no return); 5: ds_add_rtn_u32 conflict-free (the cursor of an in-LDS counting
  // sort); 6: ds_add_rtn_u32 with 8 lanes per address
  int idx = MODE == 1 ? (t * 2) & 8191 : ((MODE == 2 || MODE == 6) ? (t >> 3) : t);
  unsigned* su = reinterpret_cast<unsigned*>(s);
  unsigned got = 0;
  long long t0 = clock64();
  for (int i = 0; i < (MODE >= 7 ? 0 : iters); ++i) {
    if (MODE == 3) { s[idx] += 1.0f; }
    else if (MODE == 4) atomicAdd(&su[idx], 1u);
    else if (MODE == 5 || MODE == 6) got += atomicAdd(&su[idx], 1u);
    else atomicAdd(&s[idx], 1.0f);
    idx = (idx + 256) & 8191;
  }
  if (got == 0xffffffffu) out[0] = 1.f;      // keep the returned values alive
  if (MODE == 7 || MODE == 8) {              // 7: ds_add_u64 conflict-free (64-bit fixed point); 8: the same, 8 corners x 4 channels
    unsigned long long* s64 = reinterpret_cast<unsigned long long*>(s);      // of one "voxel" per lane: 32 adds per iteration
    int i64 = MODE == 8 ? (t * 4) & 4095 : t;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
      if (MODE == 7) { atomicAdd(&s64[i64], 3ull); i64 = (i64 + 256) & 4095; }
      else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) atomicAdd(&s64[(i64 + ((c & 1) + (c >> 1 & 1) * 9 + (c >> 2) * 81) * 4 + ch) & 4095], (unsigned long long)(i + ch));
        i64 = (i64 + 1028) & 4095;
      }
    }
  }
  __syncthreads();
  long long t1 = clock64();
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
  float a = 0.f;
  for (int i = t; i < 8192; i += 256) a += s[i];
  out[blockIdx.x * 256 + t] = a;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 4096;
  for (int mode = 0; mode < 9; ++mode) {
  for (int wgs = 256; wgs <= 512; wgs += 256) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(256), 0, 0, out, iters / 8, cyc);
      hipDeviceSynchronize();
    }
    long long h[256]; hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
    const double lanes = mode == 8 ? 256.0 * 32 * (iters / 8) : 256.0 * iters;          // lane-atomics per workgroup
    const int per_cu = mode >= 7 ? wgs / 256 : 1;
    printf("mode %d: %.0f cycles for %.0f lane-atomics per workgroup, %d workgroup(s) per CU -> %.2f lanes/clk/CU\n", mode, c, lanes,
           per_cu, per_cu * lanes / c);
    if (mode < 7) break;
  }
  }
  return 0;
}
