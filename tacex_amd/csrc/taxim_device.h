// Device-side helpers shared by the Taxim translation units (taxim_kernels.hip, taxim_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace tacex {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// compile-time loop: every index is an integral_constant, so tap indices fold to constants and the
// "tap x window" bodies become straight-line v_pk_fma_f32 streams (a plain #pragma unroll of the
// 76 x 16 nest is only partially honoured by the unroller).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// plain v_min_f32 (no NaN-canonicalising v_max in front of it; inputs are never NaN here)
__device__ __forceinline__ float fmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // torch 'reflect' (no edge repeat), single reflection: valid for -(n-1) <= i <= 2(n-1)
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave-wide inclusive scans on the VALU (DPP row shifts / broadcasts, no LDS round trips like ds_bpermute): lane 63 ends up
// with the reduction over the wave.  Sequence of LLVM's AMDGPU atomic optimizer for wave64 on gfx9.
#define TACEX_DPP_SCAN(v, IDENT, OP, T_AS_INT, INT_AS_T)                                                              \
  do {                                                                                                                \
    v = OP(v, INT_AS_T(__builtin_amdgcn_update_dpp(T_AS_INT(IDENT), T_AS_INT(v), 0x111, 0xf, 0xf, false)));           \
    v = OP(v, INT_AS_T(__builtin_amdgcn_update_dpp(T_AS_INT(IDENT), T_AS_INT(v), 0x112, 0xf, 0xf, false)));           \
    v = OP(v, INT_AS_T(__builtin_amdgcn_update_dpp(T_AS_INT(IDENT), T_AS_INT(v), 0x114, 0xf, 0xf, false)));           \
    v = OP(v, INT_AS_T(__builtin_amdgcn_update_dpp(T_AS_INT(IDENT), T_AS_INT(v), 0x118, 0xf, 0xf, false)));           \
    v = OP(v, INT_AS_T(__builtin_amdgcn_update_dpp(T_AS_INT(IDENT), T_AS_INT(v), 0x142, 0xa, 0xf, false)));           \
    v = OP(v, INT_AS_T(__builtin_amdgcn_update_dpp(T_AS_INT(IDENT), T_AS_INT(v), 0x143, 0xc, 0xf, false)));           \
  } while (0)
__device__ __forceinline__ int dpp_id_int(int x) { return x; }
__device__ __forceinline__ int dpp_add_int(int a, int b) { return a + b; }
__device__ __forceinline__ int wave_scan_add_lane63(int v) {
  TACEX_DPP_SCAN(v, 0, dpp_add_int, dpp_id_int, dpp_id_int);
  return v;
}
__device__ __forceinline__ float wave_scan_min_lane63(float v) {
  TACEX_DPP_SCAN(v, INFINITY, fminf, __float_as_int, __int_as_float);
  return v;
}
__device__ __forceinline__ float wave_scan_max_lane63(float v) {
  TACEX_DPP_SCAN(v, -INFINITY, fmaxf, __float_as_int, __int_as_float);
  return v;
}

// XCD-aware bijective remap of the linear block id (blocks b, b+8, b+16.. share an XCD / L2):
// consecutive logical ids land on the same XCD so a frame's bands share halo rows in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int nx = 8;
  int q = nblocks / nx, r = nblocks % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

struct ShadeArgs {
  const float* z;      // (B,H,W) deformed gel, mm
  const float* poly;   // (nb, nb, 24) f32: [im][id][c*6+k], padded 18 -> 24 floats (16-byte aligned rows)
  const float* bg;     // (H,W,3) f32 (NHWC copy of the background)
  const float* fx;     // (W,)
  const float* fy;     // (H,)
  float* rgb;          // (B,H,W,3)
  uint8_t* idx_out;    // (B,H,W,2) nullable
  int H, W, B, nb;
  float pixmm;         // 0.0295
  float sy, sx;        // H / calib_h, W / calib_w applied as "* H / calib_h" (TT:489-490)
  float calib_h, calib_w;
  float x_binr, y_binr;
  float gsy, gsx;              // 0.5 * H / calib_h / pixmm, 0.5 * W / calib_w / pixmm
  float inv_x_binr, inv_y_binr;
};


// plain v_max_f32 (see fmin_raw)
__device__ __forceinline__ float fmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// atan(x) for 0 <= x <= 1: x * P(x^2), P = degree-8 minimax fit of atan(sqrt z)/sqrt z on [0,1]; |error| <= 8.1e-8 rad
// evaluated in float32 (the reference's bins are 1.27e-2 rad wide; libm atanf is ~1 ulp = 6e-8 at pi/4).  One reduction
// (x -> 1/x above 1) instead of the two-step Cephes scheme: the libm atanf/atan2f + IEEE divisions made the shading
// ~400 VALU instructions / pixel; on gfx950 v_rcp/v_sqrt issue at quarter rate and v_min/v_max/v_cmp/v_cvt at half
// rate (scripts/hip_probes/valu_rates.hip), so a reduction step costs more than the extra polynomial terms.
__device__ __forceinline__ float atan_unit(float x) {
  const float z = x * x;
  float p = 0.0024566983338445425f;
  p = fmaf(p, z, -0.01440124586224556f);
  p = fmaf(p, z, 0.03978102654218674f);
  p = fmaf(p, z, -0.07234838604927063f);
  p = fmaf(p, z, 0.10498936474323273f);
  p = fmaf(p, z, -0.14161226153373718f);
  p = fmaf(p, z, 0.19985906779766083f);
  p = fmaf(p, z, -0.33332598209381104f);
  p = fmaf(p, z, 0.9999998807907104f);
  return p * x;
}

// atan(t), t >= 0
__device__ __forceinline__ float atan_pos(float t) {
  const bool big = t > 1.0f;
  const float p = atan_unit(big ? __builtin_amdgcn_rcpf(t) : t);
  return big ? 1.57079632679489662f - p : p;
}

// atan2(y, x) for finite inputs with (x, y) != (0, 0)
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  float a = atan_unit(fmin_raw(ax, ay) * __builtin_amdgcn_rcpf(fmax_raw(ax, ay)));  // in [0, pi/4]
  a = ay > ax ? 1.57079632679489662f - a : a;       // in [0, pi/2]
  a = x < 0.0f ? 3.14159265358979324f - a : a;      // in [0, pi]
  return __builtin_copysignf(a, y);
}

// gradient -> (magnitude bin, direction bin)  (TT:475-499, 243-247).  top/bot/lef/rig are the deformed-gel values (mm)
// around the CLAMPED pixel (replicate padding of the gradient maps, TT:501-502).
__device__ __forceinline__ void shade_bins(const ShadeArgs& a, float ztop, float zbot, float zlef, float zrig, int& im,
                                           int& id) {
  // z_px = -(Z / pixmm) (TT:238-239); dzdx = (z[y+1]-z[y-1])/2 * H / calib_h, dzdy likewise in x (TT:486-490):
  // folded into one scale per axis (a.gsy, a.gsx); note bot-top of -Z/pixmm = (ztop - zbot)/pixmm
  const float dzdx = (ztop - zbot) * a.gsy;
  const float dzdy = (zlef - zrig) * a.gsx;
  const float t = __builtin_amdgcn_sqrtf(fmaf(dzdx, dzdx, dzdy * dzdy));
  const float mag = atan_pos(t);
  // atan2(dzdx/t, dzdy/t) == atan2(dzdx, dzdy) for t > 0; grad_dir = 0 where t == 0 (TT:494-499)
  const float dir = t != 0.0f ? atan2_fast(dzdx, dzdy) : 0.0f;
  // floor == truncation: both arguments are >= 0 (mag >= 0, dir >= -pi; a NaN converts to 0), so only the upper clamp
  // of TT:246-247 can ever act (v_max_i32 / v_min_i32 issue at 2/3 rate)
  im = min((int)(mag * a.inv_x_binr), a.nb - 1);
  id = min((int)((dir + 3.14159274101257324f) * a.inv_y_binr), a.nb - 1);
}

// direction bin alone (the streaming tail skips the magnitude arc tangent for row segments whose gradients all fall into
// magnitude bin 0)
__device__ __forceinline__ int shade_dir_bin(const ShadeArgs& a, float dzdx, float dzdy, float t2) {
  // shade_bins tests sqrt(t2) != 0, and v_sqrt_f32 returns 0 for zero AND denormal arguments: same predicate without the sqrt
  const float dir = !(t2 < 1.17549435e-38f) ? atan2_fast(dzdx, dzdy) : 0.0f;
  return min((int)((dir + 3.14159274101257324f) * a.inv_y_binr), a.nb - 1);
}
__device__ __forceinline__ int shade_mag_bin(const ShadeArgs& a, float t2) {
  return min((int)(atan_pos(__builtin_amdgcn_sqrtf(t2)) * a.inv_x_binr), a.nb - 1);
}

// I_c = sum_k f_k(X, Y) * p_{c,k}(bin), f = [X^2, Y^2, XY, X, Y, 1]  (TT:148-157, 250-255): one FMA chain per channel.
// The table record is a 32-bit byte offset from the wave-uniform table base (no 64-bit VALU address arithmetic).
__device__ __forceinline__ void shade_poly(const ShadeArgs& a, int im, int id, float X, float Y, float out[3]) {
  const v4f* __restrict__ pc =
      reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(a.poly) + (unsigned)(im * a.nb + id) * 96u);
  const v4f c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3], c4 = pc[4];
  const float f0 = X * X, f1 = Y * Y, f2 = X * Y;
  out[0] = fmaf(f0, c0.x, fmaf(f1, c0.y, fmaf(f2, c0.z, fmaf(X, c0.w, fmaf(Y, c1.x, c1.y)))));
  out[1] = fmaf(f0, c1.z, fmaf(f1, c1.w, fmaf(f2, c2.x, fmaf(X, c2.y, fmaf(Y, c2.z, c2.w)))));
  out[2] = fmaf(f0, c3.x, fmaf(f1, c3.y, fmaf(f2, c3.z, fmaf(X, c3.w, fmaf(Y, c4.x, c4.y)))));
}

// normals -> bins -> polynomial -> + background -> clip for ONE pixel (TT:475-503, 237-258)
__device__ __forceinline__ void shade_pixel_core(const ShadeArgs& a, float ztop, float zbot, float zlef, float zrig,
                                                 int x, int y, float out[3], int& im_out, int& id_out) {
  const unsigned p = (unsigned)y * (unsigned)a.W + (unsigned)x;
  int im, id;
  shade_bins(a, ztop, zbot, zlef, zrig, im, id);
  float c[3];
  shade_poly(a, im, id, a.fx[(unsigned)x], a.fy[(unsigned)y], c);
  const float* __restrict__ bg = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.bg) + p * 12u);
  out[0] = __builtin_amdgcn_fmed3f(c[0] + bg[0], 0.0f, 1.0f);   // TT:257-258
  out[1] = __builtin_amdgcn_fmed3f(c[1] + bg[1], 0.0f, 1.0f);
  out[2] = __builtin_amdgcn_fmed3f(c[2] + bg[2], 0.0f, 1.0f);
  im_out = im;
  id_out = id;
}

__device__ __forceinline__ void shade_pixel_rgb(const ShadeArgs& a, float ztop, float zbot, float zlef, float zrig, int x,
                                                int y, float out[3]) {
  int im, id;
  shade_pixel_core(a, ztop, zbot, zlef, zrig, x, y, out, im, id);
}

// Four horizontally consecutive pixels (x % 4 == 0, x + 3 < W) of row y: the background (48 contiguous bytes) and the
// four column coordinates come in as 16-byte loads instead of 12 + 4 strided dwords per lane.
// zn[i] = {top, bottom, left, right} of pixel i.
__device__ __forceinline__ void shade_strip4_rgb(const ShadeArgs& a, const float zn[4][4], int x, int y, float out[12]) {
  const unsigned p = (unsigned)y * (unsigned)a.W + (unsigned)x;
  const v4f X4 = *reinterpret_cast<const v4f*>(a.fx + (unsigned)x);
  const float Y = a.fy[(unsigned)y];
  const v4f* __restrict__ bgp = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(a.bg) + p * 12u);
  const v4f b0 = bgp[0], b1 = bgp[1], b2 = bgp[2];
  const float bg[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int im, id;
    shade_bins(a, zn[i][0], zn[i][1], zn[i][2], zn[i][3], im, id);
    float c[3];
    shade_poly(a, im, id, X4[i], Y, c);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[3 * i + ch] = __builtin_amdgcn_fmed3f(c[ch] + bg[3 * i + ch], 0.0f, 1.0f);  // TT:257-258
  }
}

__device__ __forceinline__ void shade_pixel(const ShadeArgs& a, float ztop, float zbot, float zlef, float zrig, int x,
                                            int y, int b) {
  float c[3];
  int im, id;
  shade_pixel_core(a, ztop, zbot, zlef, zrig, x, y, c, im, id);
  const size_t p = (size_t)y * a.W + x;
  float* __restrict__ o = a.rgb + ((size_t)b * a.H * a.W + p) * 3;
  o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
  if (a.idx_out) {
    uint8_t* io = a.idx_out + ((size_t)b * a.H * a.W + p) * 2;
    io[0] = (uint8_t)im;
    io[1] = (uint8_t)id;
  }
}

}  // namespace tacex
