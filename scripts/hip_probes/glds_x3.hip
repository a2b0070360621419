// probe: global_load_lds_dwordx3 on gfx950 - LDS destination layout (wave-uniform base + lane * 12?) and per-lane global address
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* src, float* out, int shift) {
  __shared__ __attribute__((aligned(16))) float buf[2][256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) (&buf[0][0])[i] = -1.0f;
  __syncthreads();
  const float* p = src + (lane * 3 + shift) % 500;  // per-lane address, 4-byte aligned only
  __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)&buf[1][0], 12, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = buf[1][i];
}
int main() {
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)i;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 1024); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift = 0; shift < 2; ++shift) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("shift %d:", shift); for (int i = 0; i < 16; ++i) printf(" %g", r[i]); printf(" ... [189..195]:"); for (int i = 189; i < 196; ++i) printf(" %g", r[i]); printf("\n");
  }
  return 0;
}
