// probe: what does a v_cndmask_b32 cost on gfx950 when its lane mask (vcc or an SGPR pair) was written long ago?
// (valu_rates.hip measured 22.6 cycles for "cnd vcc const" against 3.0 for v_fma_f32; the streaming tail holds 40 such selects per row)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define DEF(name, pre, body)                                                         \
  __global__ __launch_bounds__(256) void name(float* out, float s) {                 \
    float a[8]; for (int k = 0; k < 8; ++k) a[k] = s + k + threadIdx.x;              \
    pre;                                                                             \
    for (int it = 0; it < ITER; ++it) {                                              \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { body; }                        \
    }                                                                                \
    float r = 0; _Pragma("unroll") for (int k = 0; k < 8; ++k) r += a[k];            \
    if (r == 12345.678f) out[threadIdx.x] = r;                                       \
  }
DEF(k_fma, , asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s)))
DEF(k_cnd_vcc, asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(s) : "vcc"), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s)))
DEF(k_cnd_sgpr, unsigned long long m; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[0]), "v"(s)), asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(s), "s"(m)))
DEF(k_bfi, unsigned mv = a[0] > s ? 0xffffffffu : 0u, asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(a[k]) : "v"(s), "v"(mv)))
// realistic mix: one select per three FMAs
DEF(k_mix_cnd, unsigned long long m; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[0]), "v"(s)),
    asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(s), "s"(m)))
DEF(k_mix_bfi, unsigned mv = a[0] > s ? 0xffffffffu : 0u,
    asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_bfi_b32 %0, %2, %1, %0" : "+v"(a[k]) : "v"(s), "v"(mv)))
DEF(k_mix_fma4, , asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s)))
// fresh mask: compare right before the select
DEF(k_cmp_cnd, , asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s) : "vcc"))
template <typename F> static void run(const char* nm, F kern, float* d, int per_body, int wg_per_cu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)wg_per_cu * ITER * 8 * per_body;
  printf("%-28s %d waves/SIMD %8.3f ms  %6.2f cycles per instruction and SIMD @2.4GHz\n", nm, wg_per_cu, ms, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  for (int w : {4, 2, 1}) {
    run("v_fma_f32", k_fma, d, 1, w); run("v_cndmask vcc (stale)", k_cnd_vcc, d, 1, w); run("v_cndmask sgpr pair (stale)", k_cnd_sgpr, d, 1, w);
    run("v_bfi vgpr mask", k_bfi, d, 1, w); run("cmp + cndmask (fresh)", k_cmp_cnd, d, 2, w);
    run("3 fma + cndmask stale", k_mix_cnd, d, 4, w); run("3 fma + bfi", k_mix_bfi, d, 4, w); run("4 fma", k_mix_fma4, d, 4, w);
  }
  return 0;
}
