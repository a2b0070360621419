// probe: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the Taxim kernels are made of, gfx950.
// Each kernel runs N iterations of 8 independent chains of one opcode; 4 waves per SIMD resident (latency hidden).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
#define DEF(name, decl, body, fin)                                                   \
  __global__ __launch_bounds__(256) void name(float* out, float s) {                 \
    decl;                                                                            \
    for (int it = 0; it < ITER; ++it) {                                              \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { body; }                        \
    }                                                                                \
    float r = 0; _Pragma("unroll") for (int k = 0; k < 8; ++k) r += fin;             \
    if (r == 12345.678f) out[threadIdx.x] = r;                                       \
  }
#define INIT float a[8]; for (int k = 0; k < 8; ++k) a[k] = s + k + threadIdx.x
#define INIT2 v2f a[8]; for (int k = 0; k < 8; ++k) a[k] = (v2f){s + k + threadIdx.x, s - k}
DEF(k_fma, INIT, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_mul, INIT, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_add, INIT, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_pkfma, INIT2, asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"((v2f){s, s})), a[k].x + a[k].y)
DEF(k_pkmul, INIT2, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"((v2f){s, s})), a[k].x + a[k].y)
DEF(k_rcp, INIT, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k])), a[k])
DEF(k_sqrt, INIT, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k])), a[k])
DEF(k_floor, INIT, asm volatile("v_floor_f32 %0, %0" : "+v"(a[k])), a[k])
DEF(k_cvt, INIT, asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k])), a[k])
DEF(k_min, INIT, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_cnd, INIT, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_cmp, INIT, asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(s) : "vcc"), a[k])
DEF(k_mov, INIT, asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(s)), a[k])
DEF(k_mad_u32, INIT, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_lshl_add, INIT, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_add_u32, INIT, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_mul_lo, INIT, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
template <typename F> static void run(const char* nm, F kern, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4;  // 4 workgroups of 4 waves per CU -> 4 waves per SIMD
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = 4.0 * ITER * 8;  // 4 waves per SIMD
  printf("%-12s %8.3f ms  %6.2f ns/instr/SIMD  = %5.2f cycles @2.4GHz\n", nm, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  run("v_fma_f32", k_fma, d); run("v_mul_f32", k_mul, d); run("v_add_f32", k_add, d);
  run("v_pk_fma", k_pkfma, d); run("v_pk_mul", k_pkmul, d);
  run("v_rcp_f32", k_rcp, d); run("v_sqrt_f32", k_sqrt, d); run("v_floor", k_floor, d); run("v_cvt_i32", k_cvt, d);
  run("v_min_f32", k_min, d); run("v_cndmask", k_cnd, d); run("v_cmp", k_cmp, d); run("v_mov", k_mov, d);
  run("v_mad_u32_u24", k_mad_u32, d); run("v_lshl_add", k_lshl_add, d); run("v_add_u32", k_add_u32, d); run("v_mul_lo_u32", k_mul_lo, d);
  return 0;
}
