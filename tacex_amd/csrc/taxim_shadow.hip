// Taxim shadow branch (with_shadow=True) for MI355X - reference semantics: taxim_torch.py:260-346 (TT).
//
//   raw shading (no background) + grad_dir                      TT:237-253, 475-503      shade_raw_kernel
//   contact mask dilated by two box rounds -> boundary ring      TT:261-273               shadow_ray_kernel
//   per ring pixel: direction bin, height bin, 4-ray fan x 51 steps of (0.625, 0.625) px; a sample is kept when it
//   is inside the image and the target is higher than the source; per-pixel/channel MIN of the table value
//   (the reference's torch_scatter.scatter_min, TT:324-337) = float atomic min                TT:275-337
//   min(shade, shadow) -> blur(shadow_blur_sigma) -> + background -> blur(deform_final_sigma) -> clip   TT:337-346
//
// The ragged (N_boundary, 4, 51) gather/scatter of the reference becomes one thread per pixel that ray-marches only
// if it is on the ring; atomics are integer min/max on the float bit pattern (table values are negative offsets).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "tacex_internal.h"
#include "taxim_device.h"

namespace tacex {

// raw polynomial shading (no background, no clip) + gradient direction; (B,H,W,3) and (B,H,W)
__global__ __launch_bounds__(256) void shade_raw_kernel(ShadeArgs a, float* __restrict__ raw, float* __restrict__ gdir) {
  const int H = a.H, W = a.W, npix = H * W;
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int y = p / W, x = p - y * W;
  const float* __restrict__ z = a.z + (size_t)b * npix;
  const int yc = min(max(y, 1), H - 2), xc = min(max(x, 1), W - 2);
  const float ztop = z[(size_t)(yc - 1) * W + xc], zbot = z[(size_t)(yc + 1) * W + xc];
  const float zlef = z[(size_t)yc * W + xc - 1], zrig = z[(size_t)yc * W + xc + 1];
  const float dzdx = (ztop - zbot) * a.gsy, dzdy = (zlef - zrig) * a.gsx;
  const float t = __builtin_amdgcn_sqrtf(dzdx * dzdx + dzdy * dzdy);
  const float mag = atan_pos(t);
  const float dir = t != 0.0f ? atan2_fast(dzdx, dzdy) : 0.0f;
  int im = (int)floorf(mag * a.inv_x_binr), id = (int)floorf((dir + 3.14159274101257324f) * a.inv_y_binr);
  im = min(max(im, 0), a.nb - 1);
  id = min(max(id, 0), a.nb - 1);
  const v4f* __restrict__ pc = reinterpret_cast<const v4f*>(a.poly + ((unsigned)(im * a.nb + id)) * 24u);
  const v4f c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3], c4 = pc[4];
  const float X = a.fx[x], Y = a.fy[y];
  const float f0 = X * X, f1 = Y * Y, f2 = X * Y;
  float* o = raw + ((size_t)b * npix + p) * 3;
  o[0] = ((((f0 * c0.x + f1 * c0.y) + f2 * c0.z) + X * c0.w) + Y * c1.x) + c1.y;
  o[1] = ((((f0 * c1.z + f1 * c1.w) + f2 * c2.x) + X * c2.y) + Y * c2.z) + c2.w;
  o[2] = ((((f0 * c3.x + f1 * c3.y) + f2 * c3.z) + X * c3.w) + Y * c4.x) + c4.y;
  gdir[(size_t)b * npix + p] = dir;
}

__global__ void fill_kernel(float* __restrict__ p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__device__ __forceinline__ void atomic_min_float(float* addr, float v) {
  if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

struct ShadowRayArgs {
  const float* z;         // (B,H,W) deformed gel, mm
  const uint8_t* mask;    // (B,H,W) shrunken contact mask
  const float* gdir;      // (B,H,W)
  const float* gel;       // (H,W)
  const float* fan_cos;   // (ndir, nfan) float32 cos / sin of the fan angles, computed on the host (see tacex_shadow_params)
  const float* fan_sin;
  const float* table;     // (ndir, nheight, nstep, 4)  [r,g,b,pad], +inf padded
  float* shadow;          // (B,H,W,3) initialised to +inf
  int H, W, B, ndir, nfan, nheight, nstep;
  int wl, wr, wt, wb;     // dilation window: a pixel is "enlarged" if any mask pixel in [x-wl, x+wr] x [y-wt, y+wb]
  float pixmm, depth0, height_prec, disc_prec, step_x, step_y;
};

__global__ __launch_bounds__(256) void shadow_ray_kernel(ShadowRayArgs a) {
  // Every float32 expression below feeds a floor / truncation to an integer (table bins, sample pixel): it must round where the
  // reference's separate torch ops round.  The library is built with -ffp-contract=fast; a fused multiply-add here moved
  // samples that sit on an integer boundary into the neighbouring pixel.
#pragma clang fp contract(off)
  const int H = a.H, W = a.W, npix = H * W;
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int y = p / W, x = p - y * W;
  const uint8_t* __restrict__ m = a.mask + (size_t)b * npix;
  if (m[p]) return;  // boundary = enlarged & ~contact (TT:272-273)
  bool enlarged = false;
  for (int yy = max(y - a.wt, 0); yy <= min(y + a.wb, H - 1) && !enlarged; ++yy)
    for (int xx = max(x - a.wl, 0); xx <= min(x + a.wr, W - 1); ++xx)
      if (m[(size_t)yy * W + xx]) { enlarged = true; break; }
  if (!enlarged) return;
  const float* __restrict__ z = a.z + (size_t)b * npix;
  const float zsrc_px = z[p] / a.pixmm;  // deformed_gel_px (TT:238)
  // direction bin (TT:276-277) and height bin (TT:280-287)
  int nidx = (int)floorf((a.gdir[(size_t)b * npix + p] + 3.14159274101257324f) / a.disc_prec);
  nidx = min(max(nidx, 0), a.ndir - 1);
  const float contact_px = (a.gel[p] - z[p]) / a.pixmm;
  int hidx = (int)floorf((contact_px * a.pixmm - a.depth0) / a.height_prec) + 6;
  const int max_h = a.nheight - 1;
  if (hidx < 0 || hidx >= max_h) hidx = max_h;
  const float* __restrict__ tab = a.table + ((size_t)nidx * a.nheight + hidx) * a.nstep * 4;
  float* __restrict__ sh = a.shadow + (size_t)b * npix * 3;
  for (int f = 0; f < a.nfan; ++f) {
    const float cs = a.fan_cos[nidx * a.nfan + f], sn = a.fan_sin[nidx * a.nfan + f];
    for (int s = 0; s < a.nstep; ++s) {
      const v4f v = *reinterpret_cast<const v4f*>(tab + s * 4);
      if (isinf(v.x) && isinf(v.y) && isinf(v.z)) continue;  // padding
      // float32, in the reference's order: (step * (s+1)) -> * cos(theta) -> x + ..., truncated toward zero like .long() (TT:298-305)
      const float tx = a.step_x * (float)(s + 1), ty = a.step_y * (float)(s + 1);
      const float ux = tx * cs, uy = ty * sn;
      const float fx = (float)x + ux, fy = (float)y + uy;
      const int sx = (int)fx, sy = (int)fy;
      if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
      if (!(zsrc_px < z[(size_t)sy * W + sx] / a.pixmm)) continue;  // TT:312-315: target must be higher
      float* o = sh + ((size_t)sy * W + sx) * 3;
      atomic_min_float(o + 0, v.x);
      atomic_min_float(o + 1, v.y);
      atomic_min_float(o + 2, v.z);
    }
  }
}

// generic small 2-D Gaussian on channels-last images with reflect borders:
//   out = clip?( blur( min?(src, other) ) + add? )
struct Blur3Args {
  const float* src; const float* other; const float* add; float* dst;
  const float* taps_w; const float* taps_h; int kw, kh;
  int H, W, B; int clip01;
};

__global__ __launch_bounds__(256) void blur_nhwc3_kernel(Blur3Args a) {
  const int H = a.H, W = a.W, npix = H * W;
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int y = p / W, x = p - y * W;
  const float* __restrict__ s = a.src + (size_t)b * npix * 3;
  const float* __restrict__ o = a.other ? a.other + (size_t)b * npix * 3 : nullptr;
  const int rw = (a.kw - 1) / 2, rh = (a.kh - 1) / 2;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  for (int j = 0; j < a.kh; ++j) {
    const int yy = reflect_idx(y - rh + j, H);
    const float wy = a.taps_h[j];
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int i = 0; i < a.kw; ++i) {
      const int xx = reflect_idx(x - rw + i, W);
      const size_t q = ((size_t)yy * W + xx) * 3;
      float v0 = s[q], v1 = s[q + 1], v2 = s[q + 2];
      if (o) { v0 = fminf(v0, o[q]); v1 = fminf(v1, o[q + 1]); v2 = fminf(v2, o[q + 2]); }
      const float wx = a.taps_w[i];
      r0 = fmaf(wx, v0, r0); r1 = fmaf(wx, v1, r1); r2 = fmaf(wx, v2, r2);
    }
    acc0 = fmaf(wy, r0, acc0); acc1 = fmaf(wy, r1, acc1); acc2 = fmaf(wy, r2, acc2);
  }
  if (a.add) { acc0 += a.add[(size_t)p * 3]; acc1 += a.add[(size_t)p * 3 + 1]; acc2 += a.add[(size_t)p * 3 + 2]; }
  if (a.clip01) {
    acc0 = fminf(fmaxf(acc0, 0.f), 1.f); acc1 = fminf(fmaxf(acc1, 0.f), 1.f); acc2 = fminf(fmaxf(acc2, 0.f), 1.f);
  }
  float* d = a.dst + ((size_t)b * npix + p) * 3;
  d[0] = acc0; d[1] = acc1; d[2] = acc2;
}

hipError_t run_shadow_rays(const ShadowParams& sw, const ShadeParams& sp, const float* z, const uint8_t* mask, const float* gel,
                           const float* gdir, float* shadow_min, int B, hipStream_t st) {
  const int H = sp.H, W = sp.W, npix = H * W;
  const dim3 grid((npix + 255) / 256, B);
  const size_t n3 = (size_t)B * npix * 3;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n3 + 255) / 256 < 65536 ? (n3 + 255) / 256 : 65536)), dim3(256), 0, st,
                     shadow_min, INFINITY, n3);
  ShadowRayArgs r{};
  r.z = z; r.mask = mask; r.gdir = gdir; r.gel = gel; r.fan_cos = sw.fan_cos_dev; r.fan_sin = sw.fan_sin_dev; r.table = sw.table_dev;
  r.shadow = shadow_min;
  r.H = H; r.W = W; r.B = B; r.ndir = sw.ndir; r.nfan = sw.nfan; r.nheight = sw.nheight; r.nstep = sw.nstep;
  r.wl = sw.wl; r.wr = sw.wr; r.wt = sw.wt; r.wb = sw.wb;
  r.pixmm = sp.pixmm; r.depth0 = sw.depth0; r.height_prec = sw.height_prec; r.disc_prec = sw.disc_prec;
  r.step_x = sw.step_x; r.step_y = sw.step_y;
  hipLaunchKernelGGL(shadow_ray_kernel, grid, dim3(256), 0, st, r);
  return hipGetLastError();
}

hipError_t run_shadow(const ShadowParams& sw, const ShadeParams& sp, const float* z, const uint8_t* mask, const float* gel,
                      float* rgb, float* ws_raw, float* ws_shadow, float* ws_gdir, float* ws_tmp, int B, hipStream_t st) {
  const int H = sp.H, W = sp.W, npix = H * W;
  ShadeArgs a{};
  a.z = z; a.poly = sp.poly_dev; a.bg = sp.bg_nhwc_dev; a.fx = sp.fx_dev; a.fy = sp.fy_dev; a.rgb = nullptr; a.idx_out = nullptr;
  a.H = H; a.W = W; a.B = B; a.nb = sp.nb; a.pixmm = sp.pixmm; a.calib_h = (float)sp.calib_h; a.calib_w = (float)sp.calib_w;
  a.x_binr = sp.x_binr; a.y_binr = sp.y_binr;
  a.gsy = (float)(0.5 * H / sp.calib_h / (double)sp.pixmm); a.gsx = (float)(0.5 * W / sp.calib_w / (double)sp.pixmm);
  a.inv_x_binr = (float)(1.0 / (double)sp.x_binr); a.inv_y_binr = (float)(1.0 / (double)sp.y_binr);
  const dim3 grid((npix + 255) / 256, B);
  hipLaunchKernelGGL(shade_raw_kernel, grid, dim3(256), 0, st, a, ws_raw, ws_gdir);
  if (hipError_t e = run_shadow_rays(sw, sp, z, mask, gel, ws_gdir, ws_shadow, B, st); e != hipSuccess) return e;
  Blur3Args b1{};
  b1.src = ws_raw; b1.other = ws_shadow; b1.add = sp.bg_nhwc_dev; b1.dst = ws_tmp;
  b1.taps_w = sw.sblur_taps_w_dev; b1.taps_h = sw.sblur_taps_h_dev; b1.kw = sw.sblur_kw; b1.kh = sw.sblur_kh;
  b1.H = H; b1.W = W; b1.B = B; b1.clip01 = 0;
  hipLaunchKernelGGL(blur_nhwc3_kernel, grid, dim3(256), 0, st, b1);
  Blur3Args b2{};
  b2.src = ws_tmp; b2.other = nullptr; b2.add = nullptr; b2.dst = rgb;
  b2.taps_w = sw.final_taps_w_dev; b2.taps_h = sw.final_taps_h_dev; b2.kw = sw.final_kw; b2.kh = sw.final_kh;
  b2.H = H; b2.W = W; b2.B = B; b2.clip01 = 1;
  hipLaunchKernelGGL(blur_nhwc3_kernel, grid, dim3(256), 0, st, b2);
  return hipGetLastError();
}

}  // namespace tacex
