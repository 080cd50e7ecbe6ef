// Does v_mfma_f32_16x16x32_f16 on gfx950 keep SUBNORMAL f16 inputs (or flush them to zero, as MI200's did)?  Decides whether a
// two-piece f16 split (x = hi + lo, |lo| <= 2^-11 |x|: subnormal in f16 for |x| < 0.125) can carry fp32 operands.
//   hipcc --offload-arch=gfx950 -O3 -o build/f16_denorm_probe tools/micro/f16_denorm_probe.hip && build/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  a[0] = (_Float16)a_val;            // every lane: A row m, k = 8 * (lane >> 4): one non-zero per lane
  b[0] = (_Float16)b_val;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
  float* d; hipMalloc(&d, 16);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {1.0f, 3.0e-5f}, {3.0e-5f, 1024.f}, {6.0e-8f, 1.0f}, {3.0e-5f, 3.0e-5f}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    // 4 k-groups (lane >> 4) each contribute a[0] * b[0] to D[0][0]  -> expected 4 * a * b
    printf("a %.3e (f16 %.6e)  b %.3e (f16 %.6e)  ->  D[0][0] = %.6e   expected %.6e\n", c[0], h[1], c[1], h[2], h[0], 4.0 * (double)h[1] * (double)h[2]);
  }
  return 0;
}
