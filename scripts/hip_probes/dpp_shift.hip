#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false)); }
__global__ void k(const float* in, float* out, float w0, float w1, float w2) {
  float a = in[threadIdx.x];
  float l = shr1(a), r = shl1(a);
  out[threadIdx.x] = w0 * l + w1 * a + w2 * r;
  out[64 + threadIdx.x] = shr1(shr1(a));
  out[128 + threadIdx.x] = shl1(shl1(a));
}
int main() {
  float h[64], *d, *o, r[192];
  for (int i = 0; i < 64; ++i) h[i] = i + 1;
  hipMalloc(&d, 256); hipMalloc(&o, 768); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 100.f, 1.f, 0.01f);
  hipMemcpy(r, o, 768, hipMemcpyDeviceToHost);
  for (int i : {0, 1, 15, 16, 17, 31, 32, 33, 62, 63}) printf("lane %d: %.2f  shr2 %.0f shl2 %.0f\n", i, r[i], r[64 + i], r[128 + i]);
  return 0;
}
