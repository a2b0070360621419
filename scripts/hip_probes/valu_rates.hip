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
DEF(k_med3f, INIT, asm volatile("v_med3_f32 %0, %0, %1, 1.0" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_med3i, INIT, asm volatile("v_med3_i32 %0, %0, %1, 0" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_maxi, INIT, asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_dpp, INIT, asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k])), a[k])
DEF(k_fma_dpp, INIT, asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_bfi, INIT, asm volatile("v_bfi_b32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_fma_sgpr, INIT, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "s"(s)), a[k])
DEF(k_fmac, INIT, asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_fmaak, INIT, asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800000" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_cmp_i, INIT, asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(a[k]), "v"(s) : "vcc"), a[k])
DEF(k_readlane, INIT, { int t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(a[k])); asm volatile("" :: "s"(t)); }, a[k])
DEF(k_and, INIT, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(s)), a[k])
DEF(k_rsq, INIT, asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k])), a[k])
DEF(k_cnd_vcc, INIT; asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(s) : "vcc"), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s) : ), a[k])
DEF(k_cmp_cnd, INIT, asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s) : "vcc"), a[k])
DEF(k_cmp_cnd_s, INIT, { unsigned long long m; asm volatile("v_cmp_gt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[k]), "=&s"(m) : "v"(s)); }, a[k])
DEF(k_cmp_fma_cnd, INIT, asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_fmac_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(s) : "vcc"), a[k])
DEF(k_cvt_f, INIT, asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k])), a[k])
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
  run("v_med3_f32", k_med3f, d); run("v_med3_i32", k_med3i, d); run("v_max_i32", k_maxi, d); run("v_mov_dpp", k_dpp, d); run("v_fmac_dpp", k_fma_dpp, d);
  run("v_bfi", k_bfi, d); run("v_fma sgpr", k_fma_sgpr, d); run("v_fmac", k_fmac, d); run("v_fmaak", k_fmaak, d); run("v_cmp_i32", k_cmp_i, d);
  run("v_readlane", k_readlane, d); run("v_and", k_and, d); run("v_rsq", k_rsq, d); run("v_cvt_f32_i32", k_cvt_f, d);
  run("cnd vcc const", k_cnd_vcc, d); run("cmp+cnd (x2)", k_cmp_cnd, d); run("cmp+cnd s (x2)", k_cmp_cnd_s, d); run("cmp,fmac,mul,cnd(x4)", k_cmp_fma_cnd, d);
  return 0;
}
