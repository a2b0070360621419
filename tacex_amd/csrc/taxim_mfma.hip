// Matrix-core (v_mfma_f32_16x16x4_f32) band kernels of the Taxim pyramid levels - see the banner below.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "tacex_internal.h"
#include "taxim_blur.h"
#include "taxim_device.h"

namespace tacex {

// ------------------------------------------------------------------------------------------------
// MFMA band kernel: the separable Gaussian as two banded-Toeplitz matrix products on the matrix cores, with
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate, bit-identical to a k-ordered fmaf chain, 32 cycles per 1024 MACs).
//   band   = NTILE x 16 rows x W columns of one frame; a 16-row tile contracts over WIN = 16 + 2 RA rows / columns,
//            RA = R rounded up to 8 (zero taps beyond R)
//   V-pass : Out[16 x 64] = Tv[16 x WIN] * In[WIN x 64] per tile and 64-column super-block (one per wave).  In comes
//            straight from global / L2: ONE 16-byte load per lane and k-step feeds 4 MFMAs of EVERY tile whose window
//            holds the row (column j of block v is column c0 + 4 j + v, so the four accumulators of a lane hold 4
//            adjacent columns -> 16-byte LDS stores).  k-step ks of lane group g = lane >> 4 contracts window row
//            4 ks + g, so tile t uses union k-steps [4 t, 4 t + KS) with the SAME weight table.  The L2 -> L1 read
//            amplification (16 NTILE + 2 RA) / (16 NTILE) is what bounds the k = 61 level: 5x at NTILE = 1, 3x at 2.
//   H-pass : Out^T[16 cols x 16 rows] = Th^T * Mid^T per 16-column block, Mid from LDS.  Here k-step ks of lane group g
//            contracts window column KS g + ks (each group walks CONSECUTIVE columns: one ds_read_b128 = four k-steps),
//            and the transposed product leaves a lane with four consecutive columns of one row, so the masked restore
//            and the 16-byte global stores run straight from the accumulators.
// Band weights per lane (0 outside the band), i = lane & 15:
//   V: wl[ks] = w[4 ks + g - RA - i + R]    H: wl[ks] = w[KS g + ks - RA - i + R]     (host tables, [2][KS][64])
// Useful MACs / issued MACs = K / WIN (0.76 at K = 61, 0.69 at 33, 0.53 at 17).
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef TACEX_MFMA_CH
#define TACEX_MFMA_CH 4
#endif
#ifndef TACEX_MFMA_FENCE
#define TACEX_MFMA_FENCE 0
#endif
// H-pass contraction map: 0 = window column KS g + ks (one ds_read_b128 per four k-steps, default), 1 = 4 ks + g like the
// V-pass (taps of every output column added in the same order; one ds_read_b32 per k-step, measured 3-5 % slower and
// without an observable difference in the bins of flat regions)
#ifndef TACEX_MFMA_H_CONSEC
#define TACEX_MFMA_H_CONSEC 0
#endif

template <int K, bool FIRST, int NTILE, bool GZ>
__global__ __launch_bounds__(640) void blur_mfma_kernel(BlurArgs a) {
  constexpr int TH = 16 * NTILE;
  constexpr int R = (K - 1) / 2, RA = (R + 7) & ~7, WIN = 16 + 2 * RA, KS = WIN / 4;
  constexpr int KU = KS + 4 * (NTILE - 1);  // k-steps over the union window of the band's tiles
#ifndef TACEX_MFMA_CH61
#define TACEX_MFMA_CH61 TACEX_MFMA_CH
#endif
#ifndef TACEX_MFMA_CH117
#define TACEX_MFMA_CH117 TACEX_MFMA_CH
#endif
  constexpr int CH = K == 61 ? TACEX_MFMA_CH61 : (K == 117 ? TACEX_MFMA_CH117 : TACEX_MFMA_CH), NCH = KU / CH;  // k-steps per software-pipeline chunk
  static_assert(KU % CH == 0 && KS % 4 == 0, "window");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* mid = reinterpret_cast<float*>(smem_raw);  // TH x pitch, columns padded by RA on both sides
  const int H = a.H, W = a.W, pitch = a.pitch;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int frame = lid / a.nbands;
  const int band = lid - frame * a.nbands;
  // the last band of a frame is shifted up when TH does not divide H: its first rows recompute (identically) what the
  // band above stores - cheaper than a second, nearly empty launch for the remainder
  const int by0 = min(a.row0 + band * TH, H - TH);
  const size_t fo = (size_t)frame * H * W;
  const float* __restrict__ src = FIRST ? a.hm + fo : a.src + fo;
  const float* __restrict__ hm = a.hm + fo;
  const float* __restrict__ gel = a.gel;
  if constexpr (GZ) {
    // Zero-band skipping: with a zero gel map the pyramid input J = min(S, 0) is non-zero only on the frame's contact rows
    // [lo, hi] (frame_rows_kernel), and level l's input only within ext_grow rows of them.  A band whose whole window lies
    // outside has an exactly zero blur and no contact pixel to restore: store zeros, read nothing.  (Reflect padding cannot
    // bring a non-zero row in: a mirrored row index -k maps to row k, which the window contains as well.)
    if (a.rows_ext != nullptr) {
      const int lo = a.rows_ext[4 * frame] - a.ext_grow, hi = a.rows_ext[4 * frame + 1] + a.ext_grow;
      if (by0 - R > hi || by0 + TH - 1 + R < lo) {
        const int w4 = W >> 2;
        for (int i = threadIdx.x; i < TH * w4; i += blockDim.x) {
          const int r = i / w4, c = i - r * w4;
          reinterpret_cast<v4f*>(a.dst + fo + (size_t)(by0 + r) * W)[c] = (v4f)(0.0f);
          if (a.mask_out) reinterpret_cast<uchar4*>(a.mask_out + fo + (size_t)(by0 + r) * W)[c] = (uchar4){0, 0, 0, 0};
        }
        return;
      }
    }
  }
  const float sa = a.shift_a[frame], sb = a.shift_b[frame];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // one 64-column super-block per wave
  const int li = lane & 15, g = lane >> 4;
  const int c0 = wid << 6;
  // Zero-BLOCK skipping (round 5): the same argument along x.  This level's input is non-zero only in the columns [in_lo, in_hi] =
  // the frame's contact columns grown by the previous levels' radii: a wave whose 64 columns lie outside has an exactly zero
  // V-pass result (it stores zeros - and zeros into the mirrored padding it owns - without loading or multiplying anything), and a
  // wave whose columns lie more than R outside has an exactly zero H-pass result and no contact pixel to restore (stores zeros,
  // reads neither the LDS rows nor the height map).  With a contact patch of ~90 columns two of the five waves of a 320-wide band
  // take these paths.  (Mirrored x-padding cannot bring a non-zero column in: position -c maps to column c, inside the window too.)
#ifndef TACEX_MFMA_BLOCK_SKIP_MIN_K
#define TACEX_MFMA_BLOCK_SKIP_MIN_K 0  // (A/B hook: levels below this kernel size are compiled without the block tests)
#endif
  bool v_need = true, h_need = true;
  if constexpr (GZ && K >= TACEX_MFMA_BLOCK_SKIP_MIN_K) {
    if (a.rows_ext != nullptr) {
      const int in_lo = a.rows_ext[4 * frame + 2] - a.ext_grow_x, in_hi = a.rows_ext[4 * frame + 3] + a.ext_grow_x;
      v_need = !(c0 + 63 < in_lo || c0 > in_hi);
      h_need = !(c0 + 63 < in_lo - R || c0 > in_hi + R);
#ifdef TACEX_MFMA_NO_BLOCK_SKIP  // (A/B hook)
      v_need = h_need = true;
#endif
    }
  }

  // ---- V-pass ----
  {
    float wl[KS];
    static_for<0, KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      wl[ks] = a.taps[ks * 64 + lane];
    });
    f32x4 acc[NTILE][4];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[t][v] = (f32x4)(0.0f);
    const int rb = by0 - RA + g;
    const unsigned coff = (unsigned)(c0 + 4 * li);
    v4f xb[2][CH], gb[FIRST ? 2 : 1][FIRST ? CH : 1];
    auto issue = [&](auto chunk_c) {
      constexpr int chunk = decltype(chunk_c)::value;
      static_for<0, CH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        int yy = reflect_idx(rb + 4 * (chunk * CH + c), H);
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);  // zero-weight rows beyond the reflect range: any finite data
        const unsigned off = (unsigned)yy * (unsigned)W + coff;
        xb[chunk & 1][c] = *reinterpret_cast<const v4f*>(src + off);
        if constexpr (FIRST) gb[chunk & 1][c] = GZ ? (v4f)(0.0f) : *reinterpret_cast<const v4f*>(gel + off);  // GZ: gel == 0 everywhere
      });
    };
    if (v_need) {
    issue(std::integral_constant<int, 0>{});
    static_for<0, NCH>([&](auto chunk_c) {
      constexpr int chunk = decltype(chunk_c)::value;
      if constexpr (TACEX_MFMA_FENCE) __builtin_amdgcn_sched_barrier(0);
      if constexpr (chunk + 1 < NCH) issue(std::integral_constant<int, chunk + 1>{});
      if constexpr (TACEX_MFMA_FENCE) __builtin_amdgcn_sched_barrier(0);
      static_for<0, CH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int ku = chunk * CH + c;
        v4f x = xb[chunk & 1][c];
        if constexpr (FIRST) {  // J = min(S, gel), S = (hm - shift_a) - shift_b (TT:441, TT:454).  fminf, not the
          const v4f gq = gb[chunk & 1][c];  // inline-asm fmin_raw: the hazard recognizer cannot see asm feeding an MFMA
          x.x = fminf((x.x - sa) - sb, gq.x); x.y = fminf((x.y - sa) - sb, gq.y);
          x.z = fminf((x.z - sa) - sb, gq.z); x.w = fminf((x.w - sa) - sb, gq.w);
        }
        static_for<0, NTILE>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int ks = ku - 4 * t;
          if constexpr (ks >= 0 && ks < KS) {
            const float w = wl[ks];
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x.x, acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x.y, acc[t][1], 0, 0, 0);
            acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x.z, acc[t][2], 0, 0, 0);
            acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x.w, acc[t][3], 0, 0, 0);
          }
        });
      });
    });
    }
    // D layout: column (lane & 15), row 4 (lane >> 4) + reg.  The lanes next to the left / right image border also write
    // the mirrored x-padding (torch 'reflect': position -c <- c, (W-1)+c <- (W-1)-c, c = 1..RA), so one barrier suffices.
    const int cx = c0 + 4 * li;
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* rowp = mid + (16 * t + 4 * g + r) * pitch + RA;
        const float e[4] = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
        *reinterpret_cast<v4f*>(rowp + cx) = (v4f){e[0], e[1], e[2], e[3]};
        if (cx <= RA) {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (cx + v >= 1 && cx + v <= RA) rowp[-(cx + v)] = e[v];
        }
        if (cx + 3 >= W - 1 - RA) {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (cx + v >= W - 1 - RA && cx + v <= W - 2) rowp[2 * (W - 1) - (cx + v)] = e[v];
        }
      }
  }
  __syncthreads();
  // ---- H-pass + masked restore: per tile, four adjacent 16-column blocks per wave (four independent chains) ----
  {
    float wl[KS];
    static_for<0, KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      wl[ks] = a.taps[((TACEX_MFMA_H_CONSEC ? 0 : KS) + ks) * 64 + lane];
    });
    const float thr = -a.pdepth[frame] * a.contact_scale;  // TT:459
    if (!h_need) {  // (wave-uniform) exactly zero output, no contact pixel in these columns
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int hrow = TACEX_MFMA_H_CONSEC ? li : mfma_h_row(li);
        const size_t p0 = (size_t)(by0 + 16 * t + hrow) * W + c0 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          *reinterpret_cast<v4f*>(a.dst + fo + p0 + 16 * n) = (v4f)(0.0f);
          if (a.mask_out) *reinterpret_cast<uchar4*>(a.mask_out + fo + p0 + 16 * n) = (uchar4){0, 0, 0, 0};
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      f32x4 acc[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = (f32x4)(0.0f);
      // restore operands: issued ahead of the MFMA chain so their latency hides behind it
      const int hrow = TACEX_MFMA_H_CONSEC ? li : mfma_h_row(li);  // tile row of this lane (see mfma_h_window_col)
      const size_t p0 = (size_t)(by0 + 16 * t + hrow) * W + c0 + 4 * g;
      v4f hv[4], gv[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        hv[n] = *reinterpret_cast<const v4f*>(hm + p0 + 16 * n);
        gv[n] = GZ ? (v4f)(0.0f) : *reinterpret_cast<const v4f*>(gel + p0 + 16 * n);
      }
      if constexpr (TACEX_MFMA_H_CONSEC) {
        // k-step ks of lane group g contracts window column 4 ks + g (same map, same table as the V-pass): every output
        // column then adds its taps in the SAME order 0..K-1, so a flat input gives a bit-exactly flat output (no
        // round-off texture for the shading to turn into noise bins).  One ds_read_b32 per block and k-step.
        const float* arow = mid + (16 * t + li) * pitch + g + c0;
        static_for<0, KS>([&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          float q[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) q[n] = arow[16 * n + 4 * ks];
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[ks], q[n], acc[n], 0, 0, 0);
        });
      } else {
        const float* arow = mid + (16 * t + hrow) * pitch + c0;  // window column mfma_h_window_col(KS, g, 4 m ..) of block n: + 16 n
        static_for<0, KS / 4>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const int hcol = mfma_h_window_col(KS, g, 4 * m);
          v4f q[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) q[n] = *reinterpret_cast<const v4f*>(arow + 16 * n + hcol);
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[4 * m], q[n].x, acc[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[4 * m + 1], q[n].y, acc[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[4 * m + 2], q[n].z, acc[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[4 * m + 3], q[n].w, acc[n], 0, 0, 0);
        });
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        v4f o;
        uint8_t mk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float S = (hv[n][k] - sa) - sb;
          const float J = fminf(S, gv[n][k]);
          const bool M = ((J - gv[n][k]) < thr) && (S < 0.0f);  // TT:457-461
          o[k] = (a.restore && M) ? J : acc[n][k];              // TT:467
          mk[k] = M ? 1 : 0;
        }
        *reinterpret_cast<v4f*>(a.dst + fo + p0 + 16 * n) = o;
        if (a.mask_out) *reinterpret_cast<uchar4*>(a.mask_out + fo + p0 + 16 * n) = (uchar4){mk[0], mk[1], mk[2], mk[3]};
      }
    }
  }
}

template <int K, bool FIRST, int NTILE, bool GZ>
static hipError_t launch_mfma_tiles(BlurArgs a, int row0, int nbands, hipStream_t st) {
  constexpr int TH = 16 * NTILE, R = (K - 1) / 2, RA = (R + 7) & ~7;
  a.row0 = row0;
  a.nbands = nbands;
  a.padx = RA;
  a.pitch = a.W + 2 * RA + 4;  // multiple of 4 with an odd quotient: the 16 rows of a ds_read_b128 hit distinct bank groups
  if (((a.pitch >> 2) & 1) == 0) a.pitch += 4;
  static const size_t lds_pad = getenv("TACEX_MFMA_LDS_PAD") ? (size_t)atoi(getenv("TACEX_MFMA_LDS_PAD")) * 1024 : 0;  // occupancy A/B hook
  const size_t lds = (size_t)TH * a.pitch * sizeof(float) + lds_pad;
  auto kern = blur_mfma_kernel<K, FIRST, NTILE, GZ>;
  static size_t granted[64] = {};
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, granted); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nbands * a.B), dim3(a.W), lds, st, a);
  return hipGetLastError();
}

// Band height: NTILE = 1 (16 rows).  Taller bands (NTILE = 2 / 3, shared V-pass loads: 3x / 2.3x instead of 5x L2 -> L1
// read amplification at k = 61) are supported by the kernel and were measured 3-8 % SLOWER at 256 x 320x240 (fewer,
// longer workgroups per CU), so only NTILE = 1 is instantiated.  a.gel == nullptr selects the all-zero-gel variant.
#ifndef TACEX_MFMA_NT117
#define TACEX_MFMA_NT117 1  // A/B: band height (16-row tiles) of the k = 117 level (640x480: 9x L2 -> L1 read amplification at 1, 5x at 2)
#endif
#ifndef TACEX_MFMA_NT61
#define TACEX_MFMA_NT61 1
#endif
template <int K, bool FIRST>
static hipError_t launch_mfma(const BlurArgs& a, hipStream_t st) {
  constexpr int NT = K == 117 ? TACEX_MFMA_NT117 : (K == 61 ? TACEX_MFMA_NT61 : 1);
  if (NT > 1 && a.H % (16 * NT) == 0 && a.W > 320) {  // (taller bands only where the window is many times the band: the 640-wide levels)
    if (a.gel == nullptr) return launch_mfma_tiles<K, FIRST, NT, true>(a, 0, a.H / (16 * NT), st);
    return launch_mfma_tiles<K, FIRST, NT, false>(a, 0, a.H / (16 * NT), st);
  }
  if (a.gel == nullptr) return launch_mfma_tiles<K, FIRST, 1, true>(a, 0, a.H / 16, st);
  return launch_mfma_tiles<K, FIRST, 1, false>(a, 0, a.H / 16, st);
}

// TACEX_BLUR_MFMA: 1 (default) = matrix-core band kernels where compiled (k = 117 / 61 / 33 / 17), 0 = VALU band kernels only
static int mfma_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("TACEX_BLUR_MFMA"); v = e ? atoi(e) : 1; }
  return v;
}

bool mfma_supported(int k, bool first, int H, int W) {
  if (!mfma_enabled() || W % 64 != 0 || W < 64 || W > 640 || H % 16 != 0) return false;
  const int R = (k - 1) / 2;
  if (R >= H || R > W - 1 || ((R + 7) & ~7) >= W) return false;
  if (first) return k == 61 || k == 117;
  return k == 117 || k == 61 || k == 33 || k == 17 || k == 15 || k == 9;
}

hipError_t dispatch_mfma(int k, bool first, const BlurArgs& a, hipStream_t st) {
  if (first) return k == 117 ? launch_mfma<117, true>(a, st) : launch_mfma<61, true>(a, st);
  switch (k) {
    case 117: return launch_mfma<117, false>(a, st);
    case 61: return launch_mfma<61, false>(a, st);
    case 33: return launch_mfma<33, false>(a, st);
    case 17: return launch_mfma<17, false>(a, st);
    case 15: return launch_mfma<15, false>(a, st);
    case 9: return launch_mfma<9, false>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace tacex
