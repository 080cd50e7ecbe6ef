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
  int idx = MODE == 1 ? (t * 2) & 8191 : (MODE == 2 ? (t >> 3) : t);
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 3) { s[idx] += 1.0f; }
    else atomicAdd(&s[idx], 1.0f);
    idx = (idx + 256) & 8191;
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
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      hipDeviceSynchronize();
    }
    long long h[256]; hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
    printf("mode %d: %.0f cycles for %d iterations of 256 lanes (1 WG/CU, 4 waves) -> %.2f lanes/clk/CU\n", mode, c, iters, 256.0 * iters / c);
  }
  return 0;
}
