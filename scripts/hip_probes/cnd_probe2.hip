// probe 2: v_cmp -> k independent FMAs -> v_cndmask ... vcc: from which distance on does the select pay the stale-VCC price?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
template <int K, bool SG>
__global__ __launch_bounds__(256) void kern(float* out, float s) {
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) { a[k] = s + k + threadIdx.x; b[k] = s * k; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsigned long long m;
      if (SG) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(s));
      else asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(s) : "vcc");
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(b[(k + j) & 7]) : "v"(s));
      if (SG) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(s), "s"(m));
      else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s) : "vcc");
    }
  }
  float r = 0;
  for (int k = 0; k < 8; ++k) r += a[k] + b[k];
  if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int K, bool SG> static void run(float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kern<K, SG>), dim3(512), dim3(256), 0, 0, d, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kern<K, SG>), dim3(512), dim3(256), 0, 0, d, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double groups = 2.0 * ITER * 8;  // per SIMD (2 waves)
  printf("cmp, %2d fma, cndmask (%s): %7.2f cycles per group = %5.2f per instruction @2.4GHz\n", K, SG ? "sgpr pair" : "vcc      ", ms * 1e6 / groups * 2.4, ms * 1e6 / groups * 2.4 / (K + 2));
}
int main() {
  float* d; hipMalloc(&d, 4096);
  run<0, false>(d); run<1, false>(d); run<2, false>(d); run<4, false>(d); run<8, false>(d); run<16, false>(d); run<32, false>(d);
  run<0, true>(d); run<4, true>(d); run<16, true>(d); run<32, true>(d);
  return 0;
}
