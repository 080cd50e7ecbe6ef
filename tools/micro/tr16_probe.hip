// Probe of ds_read_b64_tr_b16 on gfx950: which (supplier lane, element) does receiver lane L get in its element j?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tr16_probe.hip -o gpurun_out/tr16_probe && gpurun_out/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  const int l = threadIdx.x;
  for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (short)(l * 4 + e);   // lane l supplies the 8 bytes at lds + 8*l: values 4l..4l+3
  __syncthreads();
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf("  (L%2d,e%d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
