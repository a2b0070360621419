// probe 3: one v_cmp, then N selects on the same vcc / sgpr pair back to back, then 4 FMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
template <int N, bool SG>
__global__ __launch_bounds__(256) void kern(float* out, float s) {
  float a[12], b[4];
  for (int k = 0; k < 12; ++k) a[k] = s + k + threadIdx.x;
  for (int k = 0; k < 4; ++k) b[k] = s * k;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned long long m;
      if (SG) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(b[g]), "v"(s));
      else asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(b[g]), "v"(s) : "vcc");
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (SG) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(s), "s"(m));
        else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(s) : "vcc");
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(b[j]) : "v"(s));
    }
  }
  float r = 0;
  for (int k = 0; k < 12; ++k) r += a[k];
  for (int k = 0; k < 4; ++k) r += b[k];
  if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int N, bool SG> static void run(float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((kern<N, SG>), dim3(512), dim3(256), 0, 0, d, 1.0001f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((kern<N, SG>), dim3(512), dim3(256), 0, 0, d, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double groups = 2.0 * ITER * 4;
  const double cyc = ms * 1e6 / groups * 2.4;
  printf("cmp, %2d x cndmask (%s), 4 fma: %7.2f cycles per group -> %5.2f per select (after 5 x 3.0 for cmp + fma)\n", N, SG ? "sgpr pair" : "vcc      ", cyc, (cyc - 15.0) / N);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  run<1, false>(d); run<2, false>(d); run<3, false>(d); run<6, false>(d); run<12, false>(d);
  run<1, true>(d); run<3, true>(d); run<12, true>(d);
  return 0;
}
