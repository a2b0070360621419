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


// float32 arctangent on t >= 0 (Cephes atanf scheme: two-step range reduction + degree-4 odd polynomial,
// |error| ~ 1e-7 rad ~ 1 ulp).  The libm atanf/atan2f + IEEE-exact divisions made the shading VALU-bound
// (~400 instructions / pixel, 75 % of the fused kernel); bins only need the angle to ~1e-6 rad (bin width 1.27e-2),
// so results differ from libm only for pixels that sit on a bin boundary to within float roundoff.
__device__ __forceinline__ float atan_pos(float t) {
  const bool big = t > 2.414213562373095f;   // tan(3 pi / 8)
  const bool mid = t > 0.4142135623730950f;  // tan(pi / 8)
  const float x = big ? -__builtin_amdgcn_rcpf(t) : (mid ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t);
  const float y0 = big ? 1.57079632679489662f : (mid ? 0.785398163397448310f : 0.0f);
  const float z = x * x;
  const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y0 + p;
}

// atan2(y, x) for finite inputs with (x, y) != (0, 0)
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const bool swap = ay > ax;
  const float num = swap ? ax : ay, den = swap ? ay : ax;
  float a = atan_pos(num * __builtin_amdgcn_rcpf(den));        // in [0, pi/4]
  a = swap ? 1.57079632679489662f - a : a;          // in [0, pi/2]
  a = x < 0.0f ? 3.14159265358979324f - a : a;      // in [0, pi]
  return y < 0.0f ? -a : a;
}

// normals -> bins -> polynomial -> + background -> clip for ONE pixel (TT:475-503, 237-258).
// top/bot/lef/rig are the deformed-gel values (mm) around the CLAMPED pixel (replicate padding, TT:501-502).
__device__ __forceinline__ void shade_pixel_core(const ShadeArgs& a, float ztop, float zbot, float zlef, float zrig,
                                                 int x, int y, float out[3], int& im_out, int& id_out) {
  const size_t p = (size_t)y * a.W + x;
  // z_px = -(Z / pixmm) (TT:238-239); dzdx = (z[y+1]-z[y-1])/2 * H / calib_h, dzdy likewise in x (TT:486-490):
  // folded into one scale per axis (a.gsy, a.gsx); note bot-top of -Z/pixmm = (ztop - zbot)/pixmm
  const float dzdx = (ztop - zbot) * a.gsy;
  const float dzdy = (zlef - zrig) * a.gsx;
  const float t = __builtin_amdgcn_sqrtf(dzdx * dzdx + dzdy * dzdy);
  const float mag = atan_pos(t);
  // atan2(dzdx/t, dzdy/t) == atan2(dzdx, dzdy) for t > 0; grad_dir = 0 where t == 0 (TT:494-499)
  const float dir = t != 0.0f ? atan2_fast(dzdx, dzdy) : 0.0f;
  int im = (int)floorf(mag * a.inv_x_binr);                                   // TT:246
  int id = (int)floorf((dir + 3.14159274101257324f) * a.inv_y_binr);          // TT:247
  im = min(max(im, 0), a.nb - 1);
  id = min(max(id, 0), a.nb - 1);
  const v4f* __restrict__ pc = reinterpret_cast<const v4f*>(a.poly + ((unsigned)(im * a.nb + id)) * 24u);
  const v4f c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3], c4 = pc[4];
  const float X = a.fx[x], Y = a.fy[y];
  const float f0 = X * X, f1 = Y * Y, f2 = X * Y;  // TT:148-157
  // I_c = sum_k f_k * p_{c,k}
  const float r = ((((f0 * c0.x + f1 * c0.y) + f2 * c0.z) + X * c0.w) + Y * c1.x) + c1.y;
  const float g = ((((f0 * c1.z + f1 * c1.w) + f2 * c2.x) + X * c2.y) + Y * c2.z) + c2.w;
  const float bl = ((((f0 * c3.x + f1 * c3.y) + f2 * c3.z) + X * c3.w) + Y * c4.x) + c4.y;
  const float* __restrict__ bg = a.bg + p * 3;
  out[0] = fminf(fmaxf(r + bg[0], 0.0f), 1.0f);   // TT:257-258
  out[1] = fminf(fmaxf(g + bg[1], 0.0f), 1.0f);
  out[2] = fminf(fmaxf(bl + bg[2], 0.0f), 1.0f);
  im_out = im;
  id_out = id;
}

__device__ __forceinline__ void shade_pixel_rgb(const ShadeArgs& a, float ztop, float zbot, float zlef, float zrig, int x,
                                                int y, float out[3]) {
  int im, id;
  shade_pixel_core(a, ztop, zbot, zlef, zrig, x, y, out, im, id);
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
