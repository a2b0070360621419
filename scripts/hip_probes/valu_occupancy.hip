// probe: VALU issue rate of ONE SIMD as a function of the waves resident on it (gfx950): does a single wave reach the 2-cycle
// v_fma_f32 rate, or does it take several waves to fill the SIMD?  8 independent accumulators per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 8192
__global__ __launch_bounds__(64) void k_fma(float* out, float s) {
  float a[8];
  for (int k = 0; k < 8; ++k) a[k] = s + k + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s));
  }
  float r = 0;
  for (int k = 0; k < 8; ++k) r += a[k];
  if (r == 12345.678f) out[threadIdx.x] = r;
}
// the same with a scalar (SGPR) multiplier and a separate destination, like the streaming kernel's scatter chain
__global__ __launch_bounds__(64) void k_fma_s(float* out, float s) {
  float a[9];
  for (int k = 0; k < 9; ++k) a[k] = s + k + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[k]) : "s"(s), "v"(a[8]), "v"(a[k + 1]));
  }
  float r = 0;
  for (int k = 0; k < 9; ++k) r += a[k];
  if (r == 12345.678f) out[threadIdx.x] = r;
}
template <typename F> static void run(const char* nm, F kern, float* d, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4 * waves_per_simd;  // one-wave workgroups; 256 CUs x 4 SIMDs
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)waves_per_simd * ITER * 8;
  printf("%-10s waves/SIMD %d: %8.3f ms  %5.2f cycles per instruction per SIMD @2.4GHz  (%5.2f per wave)\n", nm, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd * 2.4, ms * 1e6 / (ITER * 8.0) * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  for (int w : {1, 2, 3, 4, 8}) run("v_fma", k_fma, d, w);
  for (int w : {1, 2, 4}) run("v_fma sgpr", k_fma_s, d, w);
  return 0;
}
