// Taxim optical path for MI355X (gfx950): frame-min / indentation depth, gel-pad deformation pyramid,
// polynomial shading.  Written for wave64 + LDS, no CUDA compatibility layer.
//
// Reference semantics (file:line in /root/reference, short names as in include/tacex_hip.h):
//   frame-min / indentation  TS:115-131, TT:432-441, TT:449
//   deformation pyramid      TT:443-473 (blur TT:381-412, kernels TT:362-379)
//   normals / bins / shading TT:475-503, TT:237-258, TT:139-157
//
// Data layout in HBM: height maps / deformed gel (B,H,W) f32 row-major; RGB (B,H,W,3) f32; per-frame
// scalars (B,) f32.  The blur kernel works on a band of TH rows x full width per workgroup:
//   V-pass : each thread owns 2 adjacent columns x RV rows, streams RV+K-1 rows straight from global/L2
//            (coalesced 8 B per lane) through a register sliding window;
//   LDS    : V-pass result stored row-pair interleaved [TH/2][pitch] of float2 (+ mirrored x-padding);
//   H-pass : each thread owns 2 rows x RH columns, reads its window from LDS with ds_read_b128,
//            then applies the masked restore Z[M] = J[M] and writes coalesced.
// Both passes are "tap x window" FMA loops on float2 so the compiler can use v_pk_fma_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "tacex_internal.h"
#include "taxim_device.h"
#include "taxim_blur.h"

namespace tacex {

// ------------------------------------------------------------------------------------------------
// K1: per-frame minimum (+ optional depth->mm conversion, indentation depth, uint8 camera depth)
//      one workgroup (1024 threads) per frame; float4 loads; wave shuffle + LDS reduction.
// ------------------------------------------------------------------------------------------------
template <bool FROM_DEPTH>
__global__ __launch_bounds__(1024) void frame_min_kernel(
    const float* __restrict__ in, float* __restrict__ hm_out, float* __restrict__ fmin_out,
    float* __restrict__ indent_out, uint8_t* __restrict__ cam_u8, int npix, float nmm, float far_m, float fmm,
    float gelpad_h, float gelpad_dmin) {
  const int b = blockIdx.x;
  const float* src = in + (size_t)b * npix;
  float m = INFINITY;
  const int n4 = npix >> 2;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    v4f v = reinterpret_cast<const v4f*>(src)[i];
    if (FROM_DEPTH) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = v[k];
        d = isinf(d) ? far_m : d;  // GS:585-588 (clip camera values that are inf)
        v[k] = d * 1000.0f;        // GS:590 (m -> mm)
      }
      reinterpret_cast<v4f*>(hm_out + (size_t)b * npix)[i] = v;
      if (cam_u8) {
        // GS:573-575: ((mm - near*1000) / (far*1000)) * 255 -> uint8 (divides by far, sic).  nmm / fmm are the Python
        // double products rounded once to float32 on the host (what torch does with a Python scalar operand); the
        // division is IEEE (hipcc's default correctly rounded f32 divide), so the bytes equal the reference's.
        uchar4 u;
        u.x = (uint8_t)(((v[0] - nmm) / fmm) * 255.0f);
        u.y = (uint8_t)(((v[1] - nmm) / fmm) * 255.0f);
        u.z = (uint8_t)(((v[2] - nmm) / fmm) * 255.0f);
        u.w = (uint8_t)(((v[3] - nmm) / fmm) * 255.0f);
        reinterpret_cast<uchar4*>(cam_u8 + (size_t)b * npix)[i] = u;
      }
    }
    m = fminf(m, fminf(fminf(v[0], v[1]), fminf(v[2], v[3])));
  }
  for (int i = (n4 << 2) + threadIdx.x; i < npix; i += blockDim.x) {  // tail (npix % 4)
    float d = src[i];
    if (FROM_DEPTH) {
      d = isinf(d) ? far_m : d;
      d *= 1000.0f;
      hm_out[(size_t)b * npix + i] = d;
      if (cam_u8) cam_u8[(size_t)b * npix + i] = (uint8_t)(((d - nmm) / fmm) * 255.0f);
    }
    m = fminf(m, d);
  }
  __shared__ float red[16];
  m = wave_min(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = threadIdx.x < (blockDim.x >> 6) ? red[threadIdx.x] : INFINITY;
    v = wave_min(v);
    if (threadIdx.x == 0) {
      fmin_out[b] = v;
      if (indent_out) {
        // TS:116-129
        float d = v / 1000.0f - gelpad_dmin;
        d = d < 0.0f ? 0.0f : d;
        indent_out[b] = d <= gelpad_h ? (gelpad_h - d) * 1000.0f : 0.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K1 with the CONTACT ROW RANGE of every frame as a by-product (W % 4 == 0): the same pass as frame_min_kernel, walked row
// by row (wave w takes rows w, w + 16, ...; a row is one or more float4 loads per lane), so that the per-row minima are
// available once the frame minimum is known.  S = (hm - min) - press is monotone in hm, hence a row holds a pixel with S < 0 -
// a non-zero input J = min(S, 0) of the pyramid (TT:441-454, zero gel map) - exactly when (rowmin - min) - press < 0, evaluated
// with the kernels' own expression.  rows_out[4b], rows_out[4b + 1] = first / last such row, (H, -1) when the frame has no
// contact; rows_out[4b + 2], [4b + 3] = first / last such COLUMN (round 5; (W, -1) without contact; (0, W - 1) when the block size
// cannot be a multiple of the row's float4 count).  The band kernels use the ranges to skip bands and 64-column blocks whose whole
// input window is zero (their output is exactly zero).  The block runs (1024 / (W / 4)) * (W / 4) threads so that a thread always
// meets the same four columns: their minima stay in registers and reach LDS once.
// press: indentation depth computed here (indent_out != nullptr, TS:116-129) or given per frame (press_in).
// ------------------------------------------------------------------------------------------------
constexpr int kFrameRowsMaxH = 2048;
template <bool FROM_DEPTH>
__global__ __launch_bounds__(1024) void frame_rows_kernel(
    const float* __restrict__ in, float* __restrict__ hm_out, float* __restrict__ fmin_out,
    float* __restrict__ indent_out, uint8_t* __restrict__ cam_u8, const float* __restrict__ press_in,
    int* __restrict__ rows_out, int H, int W, float nmm, float far_m, float fmm, float gelpad_h, float gelpad_dmin) {
  __shared__ float rowmin[kFrameRowsMaxH];
  __shared__ float colmin[kFrameRowsMaxH];
  __shared__ float bc[2];
  const int b = blockIdx.x;
  const size_t fo = (size_t)b * H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w4 = W >> 2;
  // The frame is walked as ONE run of float4s (thread t takes t, t + 1024, ...): every lane of every load is busy (a wave per
  // row left 48 of 128 lane slots idle at W = 320) and the iterations are independent, so several loads are in flight per
  // thread.  Row minima through LDS integer atomics on the float pattern (heights are >= 0 or +inf: the patterns order like
  // the values; a minimum does not depend on the order of its operands, so the result is the same bits as before).
  int* rowmin_i = reinterpret_cast<int*>(rowmin);
  int* colmin_i = reinterpret_cast<int*>(colmin);
  for (int r = threadIdx.x; r < H; r += blockDim.x) rowmin_i[r] = 0x7f800000;  // +inf
  const bool cols_ok = W <= kFrameRowsMaxH && (blockDim.x % w4) == 0;  // (block-uniform) a thread's float4s all lie in ONE column group
  if (cols_ok)
    for (int c = threadIdx.x; c < W; c += blockDim.x) colmin_i[c] = 0x7f800000;
  float cm[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
  __syncthreads();
  const int n4 = H * w4;
  const v4f* __restrict__ in4 = reinterpret_cast<const v4f*>(in + fo);
  // one float4 of the frame: conversion (GS:585-590), stores, the lane's minimum and its four column minima
  auto element = [&](v4f v, int q, bool valid) -> float {
    if (FROM_DEPTH) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = v[k];
        d = isinf(d) ? far_m : d;  // GS:585-588
        v[k] = d * 1000.0f;        // GS:590
      }
    }
    float m = INFINITY;
    if (valid) {
      if (FROM_DEPTH) {
        reinterpret_cast<v4f*>(hm_out + fo)[q] = v;
        if (cam_u8) {  // GS:573-575 (see frame_min_kernel)
          uchar4 c;
          c.x = (uint8_t)(((v[0] - nmm) / fmm) * 255.0f);
          c.y = (uint8_t)(((v[1] - nmm) / fmm) * 255.0f);
          c.z = (uint8_t)(((v[2] - nmm) / fmm) * 255.0f);
          c.w = (uint8_t)(((v[3] - nmm) / fmm) * 255.0f);
          reinterpret_cast<uchar4*>(cam_u8 + fo)[q] = c;
        }
      }
      m = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
      cm[0] = fminf(cm[0], v[0]); cm[1] = fminf(cm[1], v[1]); cm[2] = fminf(cm[2], v[2]); cm[3] = fminf(cm[3], v[3]);
    }
    return m;
  };
  // a wave's 64 consecutive float4s lie in at most two rows when a row has >= 64 of them (a few otherwise): reduce inside the wave
  // per row (DPP scan, the minimum lands in lane 63), one LDS atomic per row and wave
  auto row_minima = [&](float m, int r, int r_first, int r_last) {
    for (int rr = r_first; rr <= r_last; ++rr) {
      const float mr = wave_scan_min_lane63(r == rr ? m : INFINITY);
      if (lane == 63) {
        if (mr >= 0.0f) atomicMin(&rowmin_i[rr], __float_as_int(mr + 0.0f));  // (+ 0: a -0 must not pass for INT_MIN)
        else atomicMax(reinterpret_cast<unsigned*>(&rowmin_i[rr]), __float_as_uint(mr));
      }
    }
  };
  // Batches of NB float4s per thread, their loads issued back to back (clamped index instead of a predicate: one basic block).
#ifndef TACEX_FRAME_ROWS_NB
#define TACEX_FRAME_ROWS_NB 4
#endif
  constexpr int NB = TACEX_FRAME_ROWS_NB;
  if (cols_ok) {
    // The block size is a multiple of the row's float4 count: float4 q = base + t of a thread lies in row base / w4 + t / w4 and the
    // rows advance by blockDim / w4 per iteration - no division inside the loop (three of them, ~60 of its ~200 instructions
    // before; with the ds_bpermute reductions the pass was VALU-bound at 4.6 TB/s).
    const int rstep = blockDim.x / w4, rt = threadIdx.x / w4;
    const int w0 = threadIdx.x & ~63;
    const int rfw = __builtin_amdgcn_readfirstlane(w0 / w4), rlw = __builtin_amdgcn_readfirstlane((w0 + 63) / w4);
    for (int rb0 = 0; rb0 < H; rb0 += NB * rstep) {  // (wave-uniform trip counts: the wave reductions need every lane)
      v4f vb[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) vb[u] = in4[min((rb0 + u * rstep) * w4 + (int)threadIdx.x, n4 - 1)];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int rb = rb0 + u * rstep;
        if (rb >= H) break;  // (block-uniform)
        const int r = rb + rt;
        const float m = element(vb[u], rb * w4 + threadIdx.x, r < H);
        if (rb + rfw < H) row_minima(m, r < H ? r : -1, rb + rfw, min(rb + rlw, H - 1));
      }
    }
  } else {
    for (int base0 = 0; base0 < n4; base0 += NB * blockDim.x) {
      v4f vb[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) vb[u] = in4[min(base0 + u * (int)blockDim.x + (int)threadIdx.x, n4 - 1)];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int base = base0 + u * blockDim.x;
        if (base >= n4) break;  // (block-uniform)
        const int q = base + threadIdx.x;
        const bool valid = q < n4;
        const float m = element(vb[u], q, valid);
        const int q0 = base + (threadIdx.x & ~63);
        if (q0 < n4) row_minima(m, valid ? q / w4 : -1, q0 / w4, min(q0 + 63, n4 - 1) / w4);
      }
    }
  }
  if (cols_ok) {  // column minima: the same integer atomics on the float pattern as the row minima
    const int cg = (threadIdx.x % w4) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (cm[k] >= 0.0f) atomicMin(&colmin_i[cg + k], __float_as_int(cm[k] + 0.0f));
      else atomicMax(reinterpret_cast<unsigned*>(&colmin_i[cg + k]), __float_as_uint(cm[k]));
    }
  }
  __syncthreads();
  if (wave == 0) {
    float m = INFINITY;
    for (int r = lane; r < H; r += 64) m = fminf(m, rowmin[r]);
    m = wave_min(m);
    float press = 0.0f;
    if (indent_out) {  // TS:116-129
      float d = m / 1000.0f - gelpad_dmin;
      d = d < 0.0f ? 0.0f : d;
      press = d <= gelpad_h ? (gelpad_h - d) * 1000.0f : 0.0f;
    } else if (press_in) {
      press = press_in[b];
    }
    int lo = H, hi = -1;
    for (int r = lane; r < H; r += 64)
      if (((rowmin[r] - m) - press) < 0.0f) { lo = min(lo, r); hi = max(hi, r); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    int clo = 0, chi = W - 1;
    if (cols_ok) {
      clo = W; chi = -1;
      for (int c = lane; c < W; c += 64)
        if (((colmin[c] - m) - press) < 0.0f) { clo = min(clo, c); chi = max(chi, c); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { clo = min(clo, __shfl_xor(clo, o, 64)); chi = max(chi, __shfl_xor(chi, o, 64)); }
    }
    if (lane == 0) {
      fmin_out[b] = m;
      if (indent_out) indent_out[b] = press;
      rows_out[4 * b] = lo; rows_out[4 * b + 1] = hi; rows_out[4 * b + 2] = clo; rows_out[4 * b + 3] = chi;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K0 (height-map source): rasterise one analytic indenter per env straight into the height map, with the
// per-frame minimum and the indentation depth in the same pass - the input-side twin of the path.  It stands in for the
// TiledCamera depth render (GS:229-263, 581-593) when the contact geometry is a primitive: 4 B/px written, nothing read.
// Scene model (mm, pixel coordinates): gel top at gel_top_mm from the camera, background at far_clip_mm; the lowest point
// of the indenter is press_mm below the gel top; profile(x, y) = height of the indenter surface above its lowest point:
//   kind 0 sphere (radius r px)            kind 1 cylinder lying in the image plane (radius r/2, axis at angle)
//   kind 2 wedge with 45 degree flanks     kind 3 two spheres (second: centre (cx2, cy2), radius 0.6 r, 0.1 mm higher)
//   kind < 0: no contact (far clip everywhere)
// depth = min(gel_top - press + profile, far_clip).
// ------------------------------------------------------------------------------------------------
struct IndenterDesc { float kind, cx, cy, r, angle, press_mm, cx2, cy2; };

__device__ __forceinline__ float indenter_profile(const IndenterDesc& d, float sn, float cs, float x, float y, float pixmm) {
  const float dx = x - d.cx, dy = y - d.cy;
  const int kind = (int)d.kind;
  const float big = 1e3f;
  if (kind == 0 || kind == 3) {
    const float q = dx * dx + dy * dy;
    float p = q <= d.r * d.r ? (d.r - __builtin_amdgcn_sqrtf(fmaxf(d.r * d.r - q, 0.0f))) * pixmm : big;
    if (kind == 3) {
      const float r2 = 0.6f * d.r, ex = x - d.cx2, ey = y - d.cy2, q2 = ex * ex + ey * ey;
      const float p2 = q2 <= r2 * r2 ? (r2 - __builtin_amdgcn_sqrtf(fmaxf(r2 * r2 - q2, 0.0f))) * pixmm + 0.1f : big;
      p = fminf(p, p2);
    }
    return p;
  }
  const float dn = -dx * sn + dy * cs;
  if (kind == 1) {
    const float rc = 0.5f * d.r;
    return fabsf(dn) <= rc ? (rc - __builtin_amdgcn_sqrtf(fmaxf(rc * rc - dn * dn, 0.0f))) * pixmm : big;
  }
  const float dt = dx * cs + dy * sn;  // kind 2
  return (fabsf(dt) <= d.r && fabsf(dn) <= 0.6f * d.r) ? fabsf(dn) * pixmm : big;
}

__global__ __launch_bounds__(1024) void indenter_height_map_kernel(const IndenterDesc* __restrict__ desc, float* __restrict__ hm_out,
                                                                   float* __restrict__ fmin_out, float* __restrict__ indent_out,
                                                                   int H, int W, float pixmm, float gel_top_mm, float far_clip_mm,
                                                                   float gelpad_h, float gelpad_dmin) {
  const int b = blockIdx.x;
  const IndenterDesc d = desc[b];
  const float sn = sinf(d.angle), cs = cosf(d.angle);
  const float base = gel_top_mm - d.press_mm;
  const bool none = d.kind < 0.0f;
  float* out = hm_out + (size_t)b * H * W;
  float m = INFINITY;
  const int n4 = (H * W) >> 2;  // W % 4 == 0 (checked on the host): a group of 4 never straddles rows
  // (x, y) of this thread's 4-pixel group advance incrementally (no integer division per group)
  const int step_px = (int)blockDim.x << 2, step_y = step_px / W, step_x = step_px - step_y * W;
  int y = ((int)threadIdx.x << 2) / W, x = ((int)threadIdx.x << 2) - y * W;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    v4f v;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = none ? far_clip_mm : fminf(base + indenter_profile(d, sn, cs, (float)(x + k), (float)y, pixmm), far_clip_mm);
    reinterpret_cast<v4f*>(out)[i] = v;
    m = fminf(m, fminf(fminf(v[0], v[1]), fminf(v[2], v[3])));
    x += step_x; y += step_y;
    if (x >= W) { x -= W; ++y; }
  }
  __shared__ float red[16];
  m = wave_min(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = threadIdx.x < (blockDim.x >> 6) ? red[threadIdx.x] : INFINITY;
    v = wave_min(v);
    if (threadIdx.x == 0) {
      fmin_out[b] = v;
      if (indent_out) {  // TS:116-129
        float dd = v / 1000.0f - gelpad_dmin;
        dd = dd < 0.0f ? 0.0f : dd;
        indent_out[b] = dd <= gelpad_h ? (gelpad_h - dd) * 1000.0f : 0.0f;
      }
    }
  }
}

hipError_t run_indenter_height_map(const float* desc, float* hm, float* fmin, float* indent, int B, int H, int W, float pixmm,
                                   float gel_top_mm, float far_clip_mm, float gelpad_h, float gelpad_dmin, hipStream_t st) {
  hipLaunchKernelGGL(indenter_height_map_kernel, dim3(B), dim3(1024), 0, st, reinterpret_cast<const IndenterDesc*>(desc), hm, fmin,
                     indent, H, W, pixmm, gel_top_mm, far_clip_mm, gelpad_h, gelpad_dmin);
  return hipGetLastError();
}

// per-frame pressing depth P = -min(S) (TT:449) from the frame minimum:
//   shifted : S = (hm - fmin) - press  -> min(S) = (fmin - fmin) - press = -press (exact in fp32)
//   no shift: S = hm                   -> P = -fmin
__global__ void press_depth_kernel(const float* __restrict__ fmin, const float* __restrict__ press,
                                   float* __restrict__ shift_a, float* __restrict__ shift_b,
                                   float* __restrict__ pdepth, int B, int no_shift) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (no_shift) {
    shift_a[b] = 0.0f;
    shift_b[b] = 0.0f;
    pdepth[b] = -fmin[b];
  } else {
    float p = press[b];
    shift_a[b] = fmin[b];
    shift_b[b] = p;
    pdepth[b] = -((fmin[b] - fmin[b]) - p);
  }
}

// ------------------------------------------------------------------------------------------------
// K3/K4: one pyramid level = separable Gaussian + masked restore, band-tiled (see file header)
// ------------------------------------------------------------------------------------------------

template <int K, int TH, int RV, int RH, bool FIRST, int NT>
__global__ __launch_bounds__(NT, 3) void blur_band_kernel(BlurArgs a) {
  static_assert(K % 2 == 1, "odd kernel");
  static_assert(TH % RV == 0 && RV % 2 == 0 && TH % 2 == 0, "tile shape");
  constexpr int R = (K - 1) / 2;
  constexpr int RUP = (R + 1) & ~1;  // R rounded up to even (16-byte aligned LDS windows)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f* mid = reinterpret_cast<v2f*>(smem_raw);

  const int H = a.H, W = a.W;
  const int nbands = (H + TH - 1) / TH;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int frame = lid / nbands;
  const int band = lid - frame * nbands;
  const int by0 = band * TH;
  const size_t fo = (size_t)frame * H * W;
  const float* __restrict__ src = FIRST ? a.hm + fo : a.src + fo;
  const float* __restrict__ hm = a.hm + fo;
  const float* __restrict__ gel = a.gel;
  const float sa = a.shift_a[frame], sb = a.shift_b[frame];
  const float* __restrict__ taps = a.taps;
  const int pitch = a.pitch, padx = a.padx;

  // ---- V-pass: global -> registers -> LDS (row-pair interleaved) ----
  // The row chunk is WAVE-UNIFORM (waves [ch*wpc, (ch+1)*wpc) own chunk ch), so the reflected row index and the
  // row base address are scalar (SALU) work; with a per-lane chunk they cost ~8 VALU instructions per streamed row,
  // i.e. half as many again as the FMAs themselves.
  const int ncp = W >> 1;
  const int wpc = (ncp + 63) >> 6;                                        // waves per row chunk
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ch = wid / wpc;
  const int cpair = (wid - ch * wpc) * 64 + (int)(threadIdx.x & 63);
  if (ch < TH / RV && cpair < ncp) {
    const int c = cpair << 1;
    const unsigned coff = (unsigned)c * 4u;
    const int y0 = by0 + ch * RV;
    v2f acc[RV];
#pragma unroll
    for (int r = 0; r < RV; ++r) acc[r] = (v2f)(0.0f);
    static_for<0, RV + K - 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      int yy = reflect_idx(y0 - R + j, H);
      yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);  // rows of a partial last band (never stored)
      // scalar row base + 32-bit unsigned lane offset -> global_load with an SGPR base, no per-lane 64-bit adds
      const char* rowb = reinterpret_cast<const char*>(src + (size_t)yy * W);
      v2f x = *reinterpret_cast<const v2f*>(rowb + coff);
      if (FIRST) {  // J = min(S, gel), S = (hm - shift_a) - shift_b   (TT:441, TT:454)
        const char* gelb = reinterpret_cast<const char*>(gel + (size_t)yy * W);
        const v2f g = *reinterpret_cast<const v2f*>(gelb + coff);
        x.x = fmin_raw((x.x - sa) - sb, g.x);
        x.y = fmin_raw((x.y - sa) - sb, g.y);
      }
      static_for<0, RV>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int t = j - r;
        if constexpr (t >= 0 && t < K) {
          const float w = taps[t < K - 1 - t ? t : K - 1 - t];  // symmetric taps: half the scalars
          acc[r] += w * x;
        }
      });
    });
    const int rp0 = (ch * RV) >> 1;
#pragma unroll
    for (int r = 0; r < RV; r += 2) {
      v2f* row = mid + (size_t)(rp0 + (r >> 1)) * pitch + padx;
      v2f e0 = {acc[r].x, acc[r + 1].x};  // column c   : (row y, row y+1)
      v2f e1 = {acc[r].y, acc[r + 1].y};  // column c+1
      v4f q = {e0.x, e0.y, e1.x, e1.y};
      *reinterpret_cast<v4f*>(row + c) = q;
      // mirrored x-padding (torch 'reflect'): position -cc <- cc, position 2(W-1)-cc <- cc
      if (c >= 1 && c <= padx) row[-c] = e0;
      if (c + 1 <= padx) row[-(c + 1)] = e1;
      if (c >= W - 1 - padx && c <= W - 2) row[2 * (W - 1) - c] = e0;
      if (c + 1 >= W - 1 - padx && c + 1 <= W - 2) row[2 * (W - 1) - (c + 1)] = e1;
    }
  }
  __syncthreads();

  // ---- H-pass: LDS -> registers -> masked restore -> global ----
  const float P = a.pdepth[frame];
  const float thr = -P * a.contact_scale;  // TT:459
  const int nseg = W / RH;
  const int hitems = (TH >> 1) * ((nseg + 3) & ~3);
  for (int it = threadIdx.x; it < hitems; it += NT) {
    // lanes: 4 segments x 16 row pairs per wave keeps ds_read_b128 lane groups on distinct banks
    const int rp = (it >> 2) % (TH >> 1);
    const int seg = (it & 3) + ((it >> 2) / (TH >> 1)) * 4;
    if (seg >= nseg) continue;
    const int x0 = seg * RH;
    const v2f* base = mid + (size_t)rp * pitch + padx + x0 - RUP;
    v2f acc[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) acc[r] = (v2f)(0.0f);
    static_for<0, (RH + 2 * RUP) / 2>([&](auto jc) {
      constexpr int j2 = decltype(jc)::value;
      // skip 16-byte groups that carry no tap for any output (alignment slack when R is odd)
      v4f q = *reinterpret_cast<const v4f*>(base + 2 * j2);
      static_for<0, 2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        constexpr int j = 2 * j2 + h - (RUP - R);
        const v2f x = h == 0 ? (v2f){q.x, q.y} : (v2f){q.z, q.w};
        static_for<0, RH>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          constexpr int t = j - r;
          if constexpr (t >= 0 && t < K) {
            const float w = taps[t < K - 1 - t ? t : K - 1 - t];
            acc[r] += w * x;
          }
        });
      });
    });
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int y = by0 + 2 * rp + h;
      if (y >= H) continue;
      const size_t ro = (size_t)y * W + x0;
      float outv[RH];
      uint8_t mk[RH];
#pragma unroll
      for (int r4 = 0; r4 < RH; r4 += 4) {
        v4f hv = *reinterpret_cast<const v4f*>(hm + ro + r4);
        v4f gv = *reinterpret_cast<const v4f*>(gel + ro + r4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float S = (hv[k] - sa) - sb;
          const float J = fminf(S, gv[k]);
          const bool M = ((J - gv[k]) < thr) && (S < 0.0f);  // TT:457-461
          const float bl = h == 0 ? acc[r4 + k].x : acc[r4 + k].y;
          outv[r4 + k] = (a.restore && M) ? J : bl;         // TT:467
          mk[r4 + k] = M ? 1 : 0;
        }
      }
#pragma unroll
      for (int r4 = 0; r4 < RH; r4 += 4) {
        v4f o = {outv[r4], outv[r4 + 1], outv[r4 + 2], outv[r4 + 3]};
        *reinterpret_cast<v4f*>(a.dst + fo + ro + r4) = o;
      }
      if (a.mask_out) {
#pragma unroll
        for (int r4 = 0; r4 < RH; r4 += 4) {
          uchar4 u = {mk[r4], mk[r4 + 1], mk[r4 + 2], mk[r4 + 3]};
          *reinterpret_cast<uchar4*>(a.mask_out + fo + ro + r4) = u;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// looped band kernel: same band / LDS / row-pair structure as blur_band_kernel, but the tap dimension is a RUNTIME
// loop over groups of G streamed elements with zero-padded taps (wq[OFF + t] = w[t], 0 outside [0,K)).  Used where a
// full compile-time unroll is not viable (k = 117 at 640x480: 2 x 16 x 117 FMAs per thread and pass).  Cost of the
// generality: the triangular ends are computed with zero weights, (RV + K - 1 rounded up to G) / K - 1 extra FMAs
// (+13 % at k = 117).
// ------------------------------------------------------------------------------------------------
constexpr int kLoopOff = 16;  // wq index of tap 0 (>= RV - 1 + 1)

template <int TH, int RV, int RH, int G, bool FIRST, int NT>
__global__ __launch_bounds__(NT, 3) void blur_band_loop_kernel(BlurArgs a, int K) {
  static_assert(TH % RV == 0 && RV % 2 == 0 && G % 2 == 0 && RV <= kLoopOff, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f* mid = reinterpret_cast<v2f*>(smem_raw);
  const int R = (K - 1) / 2, RUP = (R + 1) & ~1;
  const int H = a.H, W = a.W;
  const int nbands = (H + TH - 1) / TH;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int frame = lid / nbands;
  const int band = lid - frame * nbands;
  const int by0 = band * TH;
  const size_t fo = (size_t)frame * H * W;
  const float* __restrict__ src = FIRST ? a.hm + fo : a.src + fo;
  const float* __restrict__ hm = a.hm + fo;
  const float* __restrict__ gel = a.gel;
  const float sa = a.shift_a[frame], sb = a.shift_b[frame];
  const float* __restrict__ wq = a.taps;  // zero-padded taps
  const int pitch = a.pitch, padx = a.padx;

  const int ncp = W >> 1;
  const int wpc = (ncp + 63) >> 6;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ch = wid / wpc;
  const int cpair = (wid - ch * wpc) * 64 + (int)(threadIdx.x & 63);
  if (ch < TH / RV && cpair < ncp) {
    const int c = cpair << 1;
    const unsigned coff = (unsigned)c * 4u;
    const int y0 = by0 + ch * RV;
    v2f acc[RV];
#pragma unroll
    for (int r = 0; r < RV; ++r) acc[r] = (v2f)(0.0f);
    const int ng = (RV + K - 1 + G - 1) / G;
#pragma unroll 1
    for (int g = 0; g < ng; ++g) {
      v2f xv[G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        int yy = reflect_idx(y0 - R + g * G + i, H);
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);  // beyond the window / partial band: finite data, zero weight
        const char* rowb = reinterpret_cast<const char*>(src + (size_t)yy * W);
        v2f x = *reinterpret_cast<const v2f*>(rowb + coff);
        if (FIRST) {
          const char* gelb = reinterpret_cast<const char*>(gel + (size_t)yy * W);
          const v2f gg = *reinterpret_cast<const v2f*>(gelb + coff);
          x.x = fmin_raw((x.x - sa) - sb, gg.x);
          x.y = fmin_raw((x.y - sa) - sb, gg.y);
        }
        xv[i] = x;
      }
      const float* wg = wq + kLoopOff + g * G;  // weight of (element i, output r) = wg[i - r]
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int r = 0; r < RV; ++r) acc[r] += wg[i - r] * xv[i];
    }
    const int rp0 = (ch * RV) >> 1;
#pragma unroll
    for (int r = 0; r < RV; r += 2) {
      v2f* row = mid + (size_t)(rp0 + (r >> 1)) * pitch + padx;
      v2f e0 = {acc[r].x, acc[r + 1].x};
      v2f e1 = {acc[r].y, acc[r + 1].y};
      v4f q = {e0.x, e0.y, e1.x, e1.y};
      *reinterpret_cast<v4f*>(row + c) = q;
      if (c >= 1 && c <= padx) row[-c] = e0;
      if (c + 1 <= padx) row[-(c + 1)] = e1;
      if (c >= W - 1 - padx && c <= W - 2) row[2 * (W - 1) - c] = e0;
      if (c + 1 >= W - 1 - padx && c + 1 <= W - 2) row[2 * (W - 1) - (c + 1)] = e1;
    }
  }
  __syncthreads();

  const float P = a.pdepth[frame];
  const float thr = -P * a.contact_scale;
  const int nseg = W / RH;
  const int hitems = (TH >> 1) * ((nseg + 3) & ~3);
  const int nwin = RH + 2 * RUP;  // float2 elements of the window
  for (int it = threadIdx.x; it < hitems; it += NT) {
    const int rp = (it >> 2) % (TH >> 1);
    const int seg = (it & 3) + ((it >> 2) / (TH >> 1)) * 4;
    if (seg >= nseg) continue;
    const int x0 = seg * RH;
    const v2f* base = mid + (size_t)rp * pitch + padx + x0 - RUP;
    v2f acc[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) acc[r] = (v2f)(0.0f);
    const int ng = (nwin + G - 1) / G;
    const int sh = RUP - R;  // 0 or 1: alignment slack in front of the window
#pragma unroll 1
    for (int g = 0; g < ng; ++g) {
      v2f xv[G];
#pragma unroll
      for (int i2 = 0; i2 < G / 2; ++i2) {
        const int j = g * G + 2 * i2;
        v4f q = (v4f)(0.0f);
        if (j < nwin) q = *reinterpret_cast<const v4f*>(base + j);  // never read beyond the row's window
        xv[2 * i2] = (v2f){q.x, q.y};
        xv[2 * i2 + 1] = (v2f){q.z, q.w};
      }
      const float* wg = wq + kLoopOff + g * G - sh;
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int r = 0; r < RH; ++r) acc[r] += wg[i - r] * xv[i];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int y = by0 + 2 * rp + h;
      if (y >= H) continue;
      const size_t ro = (size_t)y * W + x0;
      float outv[RH];
      uint8_t mk[RH];
#pragma unroll
      for (int r4 = 0; r4 < RH; r4 += 4) {
        v4f hv = *reinterpret_cast<const v4f*>(hm + ro + r4);
        v4f gv = *reinterpret_cast<const v4f*>(gel + ro + r4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float S = (hv[k] - sa) - sb;
          const float J = fminf(S, gv[k]);
          const bool M = ((J - gv[k]) < thr) && (S < 0.0f);
          const float bl = h == 0 ? acc[r4 + k].x : acc[r4 + k].y;
          outv[r4 + k] = (a.restore && M) ? J : bl;
          mk[r4 + k] = M ? 1 : 0;
        }
      }
#pragma unroll
      for (int r4 = 0; r4 < RH; r4 += 4) {
        v4f o = {outv[r4], outv[r4 + 1], outv[r4 + 2], outv[r4 + 3]};
        *reinterpret_cast<v4f*>(a.dst + fo + ro + r4) = o;
      }
      if (a.mask_out) {
#pragma unroll
        for (int r4 = 0; r4 < RH; r4 += 4) {
          uchar4 u = {mk[r4], mk[r4 + 1], mk[r4 + 2], mk[r4 + 3]};
          *reinterpret_cast<uchar4*>(a.mask_out + fo + ro + r4) = u;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// generic fallback (any odd kw, kh; any H, W): two plain passes through a temp buffer.
// Used for resolutions whose kernel sizes have no tuned instantiation (e.g. 32x32 -> (7,9)).
// ------------------------------------------------------------------------------------------------
struct GenericArgs {
  const float* src; const float* hm; const float* gel;
  const float* shift_a; const float* shift_b; const float* pdepth;
  float* dst; uint8_t* mask_out;
  const float* taps; int k;
  int H, W, B; float contact_scale; int restore; int first;
};

__global__ __launch_bounds__(256) void blur_generic_v_kernel(GenericArgs a) {  // vertical taps, src -> dst
  const int H = a.H, W = a.W;
  const size_t n = (size_t)a.B * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % W;
    const int y = (i / W) % H;
    const int b = i / ((size_t)W * H);
    const size_t fo = (size_t)b * H * W;
    const int r = (a.k - 1) / 2;
    float acc = 0.0f;
    for (int t = 0; t < a.k; ++t) {
      const int yy = reflect_idx(y - r + t, H);
      float v;
      if (a.first) {
        const float S = (a.hm[fo + (size_t)yy * W + x] - a.shift_a[b]) - a.shift_b[b];
        v = fminf(S, a.gel[(size_t)yy * W + x]);
      } else {
        v = a.src[fo + (size_t)yy * W + x];
      }
      acc = fmaf(a.taps[t], v, acc);
    }
    a.dst[i] = acc;
  }
}

__global__ __launch_bounds__(256) void blur_generic_h_kernel(GenericArgs a) {  // horizontal taps + restore
  const int H = a.H, W = a.W;
  const size_t n = (size_t)a.B * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % W;
    const int y = (i / W) % H;
    const int b = i / ((size_t)W * H);
    const size_t fo = (size_t)b * H * W;
    const int r = (a.k - 1) / 2;
    float acc = 0.0f;
    for (int t = 0; t < a.k; ++t) {
      const int xx = reflect_idx(x - r + t, W);
      acc = fmaf(a.taps[t], a.src[fo + (size_t)y * W + xx], acc);
    }
    const float g = a.gel[(size_t)y * W + x];
    const float S = (a.hm[i] - a.shift_a[b]) - a.shift_b[b];
    const float J = fminf(S, g);
    const bool M = ((J - g) < (-a.pdepth[b] * a.contact_scale)) && (S < 0.0f);
    a.dst[i] = (a.restore && M) ? J : acc;
    if (a.mask_out) a.mask_out[i] = M ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// K5-K10: normals -> bins -> polynomial gather -> + background -> clip -> NHWC   (one pass)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shade_kernel(ShadeArgs a) {
  const int H = a.H, W = a.W;
  const int npix = H * W;
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int y = p / W, x = p - y * W;
  const float* __restrict__ z = a.z + (size_t)b * npix;
  // replicate padding of the (H-2, W-2) gradient maps (TT:501-502) == evaluate at the clamped pixel
  const int yc = min(max(y, 1), H - 2), xc = min(max(x, 1), W - 2);
  shade_pixel(a, z[(size_t)(yc - 1) * W + xc], z[(size_t)(yc + 1) * W + xc], z[(size_t)yc * W + xc - 1],
              z[(size_t)yc * W + xc + 1], x, y, b);
}

// ------------------------------------------------------------------------------------------------
// K15: bilinear antialiased resize (separable triangle filter, PIL/torchvision semantics)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void aa_window(int i, float scale, int n_in, int& lo, int& cnt, float& center,
                                          float& inv) {
  const float support = scale >= 1.0f ? scale : 1.0f;
  inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
  center = scale * (i + 0.5f);
  lo = max(0, (int)(center - support + 0.5f));
  const int hi = min(n_in, (int)(center + support + 0.5f));
  cnt = hi - lo;
}

__global__ __launch_bounds__(256) void resize_aa_kernel(const float* __restrict__ src, int sh, int sw,
                                                       float* __restrict__ dst, int dh, int dw, int B, int C) {
  // channels-last: src (B,sh,sw,C) -> dst (B,dh,dw,C); C == 1 is the plain height-map case
  const size_t n = (size_t)B * dh * dw * C;
  const float scy = (float)sh / dh, scx = (float)sw / dw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % C;
    const size_t pi = i / C;
    const int x = pi % dw;
    const int y = (pi / dw) % dh;
    const int b = pi / ((size_t)dw * dh);
    int xl, xc, yl, yc;
    float cx, ix, cy, iy;
    aa_window(x, scx, sw, xl, xc, cx, ix);
    aa_window(y, scy, sh, yl, yc, cy, iy);
    float wys = 0.0f, acc = 0.0f;
    for (int jy = 0; jy < yc; ++jy) {
      const float wy = fmaxf(0.0f, 1.0f - fabsf((jy + yl - cy + 0.5f) * iy));
      float wxs = 0.0f, rowacc = 0.0f;
      const float* row = src + (((size_t)b * sh + (yl + jy)) * sw) * C + ch;
      for (int jx = 0; jx < xc; ++jx) {
        const float wx = fmaxf(0.0f, 1.0f - fabsf((jx + xl - cx + 0.5f) * ix));
        rowacc = fmaf(wx, row[(size_t)(xl + jx) * C], rowacc);
        wxs += wx;
      }
      acc = fmaf(wy, rowacc / wxs, acc);
      wys += wy;
    }
    dst[i] = acc / wys;
  }
}

// separable two-pass variant (used when a temp buffer is given): each pass has one thread per output element and
// a short contiguous tap window, so large down-sampling factors (320x240 -> 32x32 policy observation: 20 x 15
// taps) stay coalesced instead of one thread walking 300 scattered taps.
__global__ __launch_bounds__(256) void resize_aa_h_kernel(const float* __restrict__ src, int sh, int sw,
                                                         float* __restrict__ tmp, int dw, int B, int C) {
  const size_t n = (size_t)B * sh * dw * C;  // tmp (B,sh,dw,C)
  const float scx = (float)sw / dw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % C;
    const size_t pi = i / C;
    const int x = pi % dw;
    const size_t row = pi / dw;  // b*sh + y
    int xl, xc;
    float cx, ix;
    aa_window(x, scx, sw, xl, xc, cx, ix);
    const float* r = src + (row * sw) * C + ch;
    float wxs = 0.0f, acc = 0.0f;
    for (int jx = 0; jx < xc; ++jx) {
      const float wx = fmaxf(0.0f, 1.0f - fabsf((jx + xl - cx + 0.5f) * ix));
      acc = fmaf(wx, r[(size_t)(xl + jx) * C], acc);
      wxs += wx;
    }
    tmp[i] = acc / wxs;
  }
}

__global__ __launch_bounds__(256) void resize_aa_v_kernel(const float* __restrict__ tmp, int sh, float* __restrict__ dst,
                                                         int dh, int dw, int B, int C) {
  const size_t n = (size_t)B * dh * dw * C;
  const float scy = (float)sh / dh;
  const size_t rowlen = (size_t)dw * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t within = i % rowlen;
    const int y = (i / rowlen) % dh;
    const int b = i / (rowlen * dh);
    int yl, yc;
    float cy, iy;
    aa_window(y, scy, sh, yl, yc, cy, iy);
    float wys = 0.0f, acc = 0.0f;
    for (int jy = 0; jy < yc; ++jy) {
      const float wy = fmaxf(0.0f, 1.0f - fabsf((jy + yl - cy + 0.5f) * iy));
      acc = fmaf(wy, tmp[((size_t)b * sh + yl + jy) * rowlen + within], acc);
      wys += wy;
    }
    dst[i] = acc / wys;
  }
}

// float4 variant (row length a multiple of 4): 16 B per lane, taps unrolled by 4 so several rows are in flight -
// this pass streams the whole source image once and should run at HBM speed
__global__ __launch_bounds__(256) void resize_aa_v4_kernel(const float* __restrict__ src, int sh, float* __restrict__ dst,
                                                          int dh, int rowlen4, int B) {
  const size_t n = (size_t)B * dh * rowlen4;
  const float scy = (float)sh / dh;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int within = i % rowlen4;
    const int y = (i / rowlen4) % dh;
    const int b = i / ((size_t)rowlen4 * dh);
    int yl, yc;
    float cy, iy;
    aa_window(y, scy, sh, yl, yc, cy, iy);
    const v4f* p = reinterpret_cast<const v4f*>(src) + ((size_t)b * sh + yl) * rowlen4 + within;
    v4f acc = (v4f)(0.0f);
    float wys = 0.0f;
    int jy = 0;
    for (; jy + 4 <= yc; jy += 4) {
      const v4f q0 = p[(size_t)(jy + 0) * rowlen4], q1 = p[(size_t)(jy + 1) * rowlen4];
      const v4f q2 = p[(size_t)(jy + 2) * rowlen4], q3 = p[(size_t)(jy + 3) * rowlen4];
      const float w0 = fmaxf(0.0f, 1.0f - fabsf((jy + 0 + yl - cy + 0.5f) * iy));
      const float w1 = fmaxf(0.0f, 1.0f - fabsf((jy + 1 + yl - cy + 0.5f) * iy));
      const float w2 = fmaxf(0.0f, 1.0f - fabsf((jy + 2 + yl - cy + 0.5f) * iy));
      const float w3 = fmaxf(0.0f, 1.0f - fabsf((jy + 3 + yl - cy + 0.5f) * iy));
      acc += w0 * q0; acc += w1 * q1; acc += w2 * q2; acc += w3 * q3;
      wys += w0; wys += w1; wys += w2; wys += w3;
    }
    for (; jy < yc; ++jy) {
      const float w = fmaxf(0.0f, 1.0f - fabsf((jy + yl - cy + 0.5f) * iy));
      acc += w * p[(size_t)jy * rowlen4];
      wys += w;
    }
    reinterpret_cast<v4f*>(dst)[i] = acc / wys;
  }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline int lds_pitch_for(int W, int padx) {
  int pitch = W + 2 * padx;
  int m = ((2 - pitch) % 32 + 32) % 32;  // pitch == 2 (mod 32) float2 units -> row-pair stride of 4 banks
  return pitch + m;
}

template <int K, int TH, int RV, int RH, bool FIRST, int NT>
static hipError_t launch_band_nt(const BlurArgs& a0, hipStream_t st) {
  BlurArgs a = a0;
  constexpr int R = (K - 1) / 2;
  a.padx = (R + 1) & ~1;
  a.pitch = lds_pitch_for(a.W, a.padx);
  const int nbands = (a.H + TH - 1) / TH;
  const size_t lds = (size_t)(TH / 2) * a.pitch * sizeof(v2f);
  auto kern = blur_band_kernel<K, TH, RV, RH, FIRST, NT>;
  static size_t granted[64] = {};
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, granted); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nbands * a.B), dim3(NT), lds, st, a);
  return hipGetLastError();
}

// threads per workgroup: 64 * ceil(W/128) waves per row chunk * (TH/RV = 2) chunks
template <int K, int TH, int RV, int RH, bool FIRST>
static hipError_t launch_band(const BlurArgs& a, hipStream_t st) {
  // widths up to 384 only: wider frames are served by the MFMA band kernels (taxim_mfma.hip) or, with TACEX_BLUR_MFMA=0,
  // by the looped / generic kernels - the fully unrolled k = 61 / 33 instantiations for 640 columns cost a minute of build
  // time for a path nothing selects by default
  if (a.W <= 384) return launch_band_nt<K, TH, RV, RH, FIRST, 384>(a, st);
  return hipErrorInvalidValue;
}

template <bool FIRST, int NT>
static hipError_t launch_band_loop(const BlurArgs& a0, int K, hipStream_t st) {
  constexpr int TH = 32, RV = 16, RH = 16, G = 12;
  BlurArgs a = a0;
  const int R = (K - 1) / 2;
  a.padx = (R + 1) & ~1;
  a.pitch = lds_pitch_for(a.W, a.padx);
  const int nbands = (a.H + TH - 1) / TH;
  const size_t lds = (size_t)(TH / 2) * a.pitch * sizeof(v2f);
  auto kern = blur_band_loop_kernel<TH, RV, RH, G, FIRST, NT>;
  static size_t granted[64] = {};
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, granted); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nbands * a.B), dim3(NT), lds, st, a, K);
  return hipGetLastError();
}

// looped variant: any odd K with identical w/h taps whose mirrored padding fits the row (image >= 2 * radius)
static int band_loop_min_k() {  // A/B hook: TACEX_BAND_LOOP_MIN_K overrides the smallest k served by the looped kernel
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("TACEX_BAND_LOOP_MIN_K");
    v = e ? atoi(e) : 35;
  }
  return v;
}

static bool band_loop_supported(int k, int H, int W) {
  const int R = (k - 1) / 2;
  return W % 16 == 0 && W >= 32 && W <= 640 && R < H && ((R + 1) & ~1) <= W - 1 && k >= band_loop_min_k();
}

static bool band_supported(int k, int H, int W) {
  if (W % 16 != 0 || W < 32 || H < 2) return false;
  if (W > 384) return false;
  if ((k - 1) / 2 >= H || ((k - 1) / 2 + 1) > W - 1) return false;
  switch (k) {
    case 3: case 5: case 9: case 15: case 17: case 33: case 61: return true;  // 117: generic path (TODO chunked variant)
    default: return false;
  }
}

static bool band_first_supported(int k) { return k == 61; }

static hipError_t dispatch_band(int k, bool first, const BlurArgs& a, hipStream_t st) {
  if (first) {
    switch (k) {  // first pyramid level at the two tuned resolutions (320x240 / 640x480)
      case 61: return launch_band<61, 32, 16, 16, true>(a, st);
      default: return hipErrorInvalidValue;
    }
  }
  switch (k) {
    case 3: return launch_band<3, 32, 16, 16, false>(a, st);
    case 5: return launch_band<5, 32, 16, 16, false>(a, st);
    case 9: return launch_band<9, 32, 16, 16, false>(a, st);
    case 15: return launch_band<15, 32, 16, 16, false>(a, st);
    case 17: return launch_band<17, 32, 16, 16, false>(a, st);
    case 33: return launch_band<33, 32, 16, 16, false>(a, st);
    case 61: return launch_band<61, 32, 16, 16, false>(a, st);
    default: return hipErrorInvalidValue;
  }
}

// min (+ conversion / indentation) AND the contact row range of every frame; false when the geometry is not supported
// (the caller then runs run_frame_min and treats every row as contact)
bool frame_rows_supported(int H, int W) { return (W % 4) == 0 && H <= kFrameRowsMaxH; }
__global__ void fill_rows_kernel(int* rows, int B, int H, int W) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { rows[4 * b] = 0; rows[4 * b + 1] = H - 1; rows[4 * b + 2] = 0; rows[4 * b + 3] = W - 1; }
}
hipError_t run_fill_rows(int* rows, int B, int H, int W, hipStream_t st) {  // "every row / column may hold contact" (geometry without a row kernel)
  hipLaunchKernelGGL(fill_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, st, rows, B, H, W);
  return hipGetLastError();
}
hipError_t run_frame_rows(const float* in, bool from_depth, float* hm_out, float* fmin, float* indent, uint8_t* cam_u8,
                          const float* press_in, int* rows_out, int B, int H, int W, float near_mm, float far_m, float far_mm,
                          float gelpad_h, float gelpad_dmin, hipStream_t st) {
  const int w4 = W >> 2;
  int nt = w4 <= 1024 ? (1024 / w4) * w4 : 1024;  // a multiple of the row's float4 count: every thread keeps to four columns
  if (nt < 512 || (nt & 63)) nt = 1024;           // (whole waves, enough of them; else the kernel reports every column as contact)
  if (from_depth)
    hipLaunchKernelGGL(frame_rows_kernel<true>, dim3(B), dim3(nt), 0, st, in, hm_out, fmin, indent, cam_u8, press_in, rows_out,
                       H, W, near_mm, far_m, far_mm, gelpad_h, gelpad_dmin);
  else
    hipLaunchKernelGGL(frame_rows_kernel<false>, dim3(B), dim3(nt), 0, st, in, hm_out, fmin, indent, cam_u8, press_in, rows_out,
                       H, W, near_mm, far_m, far_mm, gelpad_h, gelpad_dmin);
  return hipGetLastError();
}

hipError_t run_frame_min(const float* in, bool from_depth, float* hm_out, float* fmin, float* indent,
                         uint8_t* cam_u8, int B, int npix, float near_mm, float far_m, float far_mm, float gelpad_h,
                         float gelpad_dmin, hipStream_t st) {
  if (from_depth)
    hipLaunchKernelGGL(frame_min_kernel<true>, dim3(B), dim3(1024), 0, st, in, hm_out, fmin, indent, cam_u8,
                       npix, near_mm, far_m, far_mm, gelpad_h, gelpad_dmin);
  else
    hipLaunchKernelGGL(frame_min_kernel<false>, dim3(B), dim3(1024), 0, st, in, hm_out, fmin, indent, cam_u8,
                       npix, near_mm, far_m, far_mm, gelpad_h, gelpad_dmin);
  return hipGetLastError();
}

hipError_t run_press_depth(const float* fmin, const float* press, float* sa, float* sb, float* pd, int B,
                           int no_shift, hipStream_t st) {
  hipLaunchKernelGGL(press_depth_kernel, dim3((B + 255) / 256), dim3(256), 0, st, fmin, press, sa, sb, pd, B,
                     no_shift);
  return hipGetLastError();
}

// true: run_blur_level runs this level as ONE matrix-core launch that never touches `tmp` (launches of different frame ranges may overlap)
bool blur_level_single_kernel(const LevelDesc& lv, bool first, int H, int W) {
  return lv.same_taps && lv.taps_mfma_dev && mfma_supported(lv.kw, first, H, W);
}

hipError_t run_blur_level(const LevelDesc& lv, const float* src, const float* hm, const float* gel,
                          const float* sa, const float* sb, const float* pd, float* dst, float* tmp,
                          uint8_t* mask_out, int B, int H, int W, float contact_scale, int restore,
                          bool first, hipStream_t st, const int* rows_ext, int ext_grow, int ext_grow_x) {
  if (lv.same_taps && lv.taps_mfma_dev && mfma_supported(lv.kw, first, H, W)) {
    BlurArgs a{};
    a.src = src; a.hm = hm; a.shift_a = sa; a.shift_b = sb; a.pdepth = pd;
    a.rows_ext = lv.gel_zero ? rows_ext : nullptr; a.ext_grow = ext_grow; a.ext_grow_x = ext_grow_x;  // zero-band / zero-block skipping needs J = min(S, 0)
    a.gel = lv.gel_zero ? nullptr : gel;  // all-zero gel map (GelSight Mini): no gel loads - half of level 0's V-pass reads
    a.dst = dst; a.mask_out = mask_out; a.taps = lv.taps_mfma_dev; a.H = H; a.W = W; a.B = B;
    a.contact_scale = contact_scale; a.restore = restore;
    return dispatch_mfma(lv.kw, first, a, st);
  }
  const bool unrolled_ok = band_supported(lv.kw, H, W) && (!first || band_first_supported(lv.kw));
  if (lv.same_taps && lv.taps_pad_dev && band_loop_supported(lv.kw, H, W) && (!unrolled_ok || lv.kw >= band_loop_min_k()) &&
      (band_loop_min_k() != 35 || !unrolled_ok)) {
    BlurArgs a{};
    a.src = src; a.hm = hm; a.gel = gel; a.shift_a = sa; a.shift_b = sb; a.pdepth = pd;
    a.dst = dst; a.mask_out = mask_out; a.taps = lv.taps_pad_dev; a.H = H; a.W = W; a.B = B;
    a.contact_scale = contact_scale; a.restore = restore;
    if (W <= 384) return first ? launch_band_loop<true, 384>(a, lv.kw, st) : launch_band_loop<false, 384>(a, lv.kw, st);
    return first ? launch_band_loop<true, 640>(a, lv.kw, st) : launch_band_loop<false, 640>(a, lv.kw, st);
  }
  if (lv.same_taps && band_supported(lv.kw, H, W) && (!first || band_first_supported(lv.kw))) {
    BlurArgs a{};
    a.src = src; a.hm = hm; a.gel = gel; a.shift_a = sa; a.shift_b = sb; a.pdepth = pd;
    a.dst = dst; a.mask_out = mask_out; a.taps = lv.taps_w_dev; a.H = H; a.W = W; a.B = B;
    a.contact_scale = contact_scale; a.restore = restore;
    return dispatch_band(lv.kw, first, a, st);
  }
  GenericArgs g{};
  g.hm = hm; g.gel = gel; g.shift_a = sa; g.shift_b = sb; g.pdepth = pd; g.H = H; g.W = W; g.B = B;
  g.contact_scale = contact_scale;
  const size_t n = (size_t)B * H * W;
  const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  g.src = src; g.dst = tmp; g.taps = lv.taps_h_dev; g.k = lv.kh; g.first = first ? 1 : 0; g.restore = 0;
  g.mask_out = nullptr;
  hipLaunchKernelGGL(blur_generic_v_kernel, dim3(grid), dim3(256), 0, st, g);
  g.src = tmp; g.dst = dst; g.taps = lv.taps_w_dev; g.k = lv.kw; g.first = 0; g.restore = restore;
  g.mask_out = mask_out;
  hipLaunchKernelGGL(blur_generic_h_kernel, dim3(grid), dim3(256), 0, st, g);
  return hipGetLastError();
}

hipError_t run_shade(const ShadeParams& sp, const float* z, float* rgb, uint8_t* idx_out, int B,
                     hipStream_t st) {
  ShadeArgs a{};
  a.z = z; a.poly = sp.poly_dev; a.bg = sp.bg_nhwc_dev; a.fx = sp.fx_dev; a.fy = sp.fy_dev; a.rgb = rgb;
  a.idx_out = idx_out; a.H = sp.H; a.W = sp.W; a.B = B; a.nb = sp.nb; a.pixmm = sp.pixmm;
  a.calib_h = (float)sp.calib_h; a.calib_w = (float)sp.calib_w;
  a.x_binr = sp.x_binr; a.y_binr = sp.y_binr;
  a.gsy = (float)(0.5 * sp.H / sp.calib_h / (double)sp.pixmm); a.gsx = (float)(0.5 * sp.W / sp.calib_w / (double)sp.pixmm);
  a.inv_x_binr = (float)(1.0 / (double)sp.x_binr); a.inv_y_binr = (float)(1.0 / (double)sp.y_binr);
  const int npix = sp.H * sp.W;
  hipLaunchKernelGGL(shade_kernel, dim3((npix + 255) / 256, B), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t run_resize_aa(const float* src, int sh, int sw, float* dst, int dh, int dw, int B, int C, float* tmp,
                         hipStream_t st) {
  if (tmp) {
    // vertical pass first: adjacent lanes read adjacent addresses (fully coalesced) and the data shrinks by sh/dh
    // before the channel-strided horizontal pass; tmp is (B, dh, sw, C)
    const size_t n1 = (size_t)B * dh * sw * C, n2 = (size_t)B * dh * dw * C;
    const int g1 = (int)((n1 + 255) / 256 < 65536 ? (n1 + 255) / 256 : 65536);
    const int g2 = (int)((n2 + 255) / 256 < 65536 ? (n2 + 255) / 256 : 65536);
    if ((sw * C) % 4 == 0) {
      const size_t n4 = n1 / 4;
      const int g4 = (int)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
      hipLaunchKernelGGL(resize_aa_v4_kernel, dim3(g4), dim3(256), 0, st, src, sh, tmp, dh, sw * C / 4, B);
    } else {
      hipLaunchKernelGGL(resize_aa_v_kernel, dim3(g1), dim3(256), 0, st, src, sh, tmp, dh, sw, B, C);
    }
    hipLaunchKernelGGL(resize_aa_h_kernel, dim3(g2), dim3(256), 0, st, tmp, dh, sw, dst, dw, B, C);
    return hipGetLastError();
  }
  const size_t n = (size_t)B * dh * dw * C;
  const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  hipLaunchKernelGGL(resize_aa_kernel, dim3(grid), dim3(256), 0, st, src, sh, sw, dst, dh, dw, B, C);
  return hipGetLastError();
}

}  // namespace tacex
