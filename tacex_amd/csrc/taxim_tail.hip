// Fused tail of the Taxim path: the small-kernel pyramid levels (e.g. k = 9, 5, 3 + final 5 at 320x240) with their
// masked restores AND the shading, in ONE LDS-tiled kernel.
//
// Why: as separate band kernels these levels are memory-bound (12 B/px each at ~4.5 TB/s) and the shade kernel
// re-reads the deformed gel; fused, a 64x32 output tile (+ halo = sum of radii + 2) lives in LDS across all levels,
// HBM sees one read of the previous level + the height map and one write of RGB (+ deformed gel / mask for FOTS).
//
// Reference semantics per level: Z = G(Z); Z[M] = J[M]  (TT:464-467), final Z = G(Z) without restore (TT:468-471),
// reflect padding at the IMAGE border at every level (TT:411) - reproduced by loading input, J and M at reflected
// coordinates (a symmetric blur keeps the halo mirror-symmetric); then normals/bins/polynomial/background/clip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "tacex_internal.h"
#include "taxim_device.h"

namespace tacex {

constexpr int kTailMaxLevels = 5;
// Every barrier-separated phase of the tail costs ~2 k cycles per tile whatever its work (in-kernel clock64), so the small
// levels trade FMAs for phases: k <= 5 levels compute the horizontal sums of their (2 + 2R) input rows per item in
// registers and go straight to the vertical sums - same arithmetic, same order, bit-identical results, one phase less.
#ifndef TACEX_TAIL_FUSE2D_MAXK
#define TACEX_TAIL_FUSE2D_MAXK 5
#endif
constexpr int kTailFuse2dMaxK = TACEX_TAIL_FUSE2D_MAXK;
// 8 waves per workgroup, 2 per SIMD: two workgroups per CU (LDS-bound) then need 4 wave slots and <= 128 VGPRs per SIMD.
// (640 threads measured 25 % slower: 3+3+2+2 waves per workgroup leave no room for the second workgroup's 3 on a SIMD.)
constexpr int kTailThreads = 512;
// true: J = min(S, gel) and the contact mask of the whole region stay in LDS across the three restores (28 KB of the
// 80.9 KB: TWO workgroups per CU).  false: J is not cached and the mask is kept as 4 bits per 4-pixel group (54.1 KB =
// three workgroups per CU, needs <= 80 VGPRs); the restore then re-reads hm / gel (L2 hits) for groups with a mask bit.
// Measured at 256 x 320x240 (round 1): cached 253 us; uncached at the 128-VGPR budget (still 2 per CU) 264 us; uncached
// at the 80-VGPR budget (3 per CU, 31 spilled VGPRs) 329 us - the third workgroup does not pay while the epilogue
// spills, so the cached layout stays the default.
constexpr bool kTailCacheJ = true;
constexpr int kTailWavesPerSimd = kTailCacheJ ? 4 : 6;  // register budget: 128 / 80 VGPRs  // register budget: 128 / 80 VGPRs
// observation cells a 64 x 32 tile can overlap when the down-sampling factor is >= 7.5 (y) / >= 8 (x): tile / scale + 3
// (per tail instantiation: TailCfg::ONRY / ONCX; 8 x 11 for the 320x240 tail, 6 x 7 for the 640x480 one whose filters are longer)
// longest triangle filter the fused observation handles (taps per output row / column); longer ones take the two-pass resize
// (TailCfg::OKY / OKX: 16 / 24 taps at 320x240 -> 32x32, 32 / 44 at 640x480 -> 32x32)

struct TailArgs {
  const float* zin;      // (B,H,W) output of the last band-kernel level
  const float* hm;       // (B,H,W)
  const float* gel;      // (H,W)
  const float* shift_a;  // (B,)
  const float* shift_b;
  const float* pdepth;
  const float* taps[kTailMaxLevels];
  float* z_out;          // (B,H,W) nullable
  uint8_t* mask_out;     // (B,H,W) nullable
  ShadeArgs sh;          // sh.rgb == nullptr -> deformation only
  int H, W, B;
  float contact_scale;
  float* obs_part;       // nullable: per-tile partial sums of the policy observation, [tile][ONRY][ONCX][3]
  ObsTables obs;         // filter tables of the observation (valid when obs_part != nullptr)
  FotsReduce* fots_part; // nullable: per-wave contact statistics, [tile][kTailWavesPerTile]
  // nullable: deformed gel / contact mask at the FOTS marker pixels only (B, M) - lets the caller drop the full-frame
  // z_out / mask_out stores.  mk_tile[tile_in_frame][kTailMaxMarkersPerTile] = marker << 16 | ly << 8 | lx, mk_cnt[tile]
  float* pix_z; uint8_t* pix_m; const int* mk_tile; const int* mk_cnt; int n_markers;
};

template <int... KS>
struct TailCfg {
  static constexpr int NL = sizeof...(KS);
  static constexpr int K[NL] = {KS...};
  static constexpr int sum_r() { int s = 0; for (int i = 0; i < NL; ++i) s += (K[i] - 1) / 2; return s; }
  static constexpr int max_r() { int m = 0; for (int i = 0; i < NL; ++i) m = (K[i] - 1) / 2 > m ? (K[i] - 1) / 2 : m; return m; }
  static constexpr int TW = 64, TH = 32;
  static constexpr int HLY = sum_r() + 2;            // +1 central difference, +1 clamped (replicate) gradient row
  static constexpr int HLX = (HLY + 1) & ~1;         // even -> region width stays a multiple of 4
  static constexpr int RW = TW + 2 * HLX;            // region width  (multiple of 4)
  static constexpr int RH = (TH + 2 * HLY + 3) & ~3; // region height (multiple of 4)
  static constexpr int PADX = (max_r() + 3) & ~3;    // guard columns for the 16-byte window reads
  static constexpr int PADY = max_r();               // guard rows for the vertical window
  static constexpr int P = RW + 2 * PADX + 4;        // LDS pitch (floats); +4 staggers rows across banks
  static constexpr int ROWS = RH + 2 * PADY;
  // valid margin needed after level l = sum of the radii of levels l+1.. plus 2 (gradient + replicate clamp)
  // levels with k <= kTailFuse2dMaxK run H and V in ONE phase (horizontal sums of the 2 + 2R rows an item needs stay in
  // registers): output goes to the OTHER ping-pong buffer, so the buffer holding level l's input alternates
  static constexpr bool fused2d(int l) { return K[l] > 1 && K[l] <= kTailFuse2dMaxK; }
  static constexpr int in_buf(int l) { int n = 0; for (int i = 0; i < l; ++i) n += fused2d(i) ? 1 : 0; return n & 1; }
  static constexpr int margin_after(int l) { int s = 2; for (int i = l + 1; i < NL; ++i) s += (K[i] - 1) / 2; return s; }
  // fused policy observation: cells a tile may overlap (ONRY x ONCX) and longest filter (OKY / OKX taps) this instantiation
  // is compiled for.  The three-level tail is the 640x480 one: longer filters, fewer cells per tile; sized so that the
  // tap windows still fit beside two workgroups' buffers in the CU's LDS.
  static constexpr bool wide_obs = (NL == 3 && K[0] == 9);
  static constexpr int ONRY = wide_obs ? 6 : 8, ONCX = wide_obs ? 7 : 11, OKY = wide_obs ? 32 : 16, OKX = wide_obs ? 44 : 24;
  // the fused observation stages the tile's RGB (TH x TW x 3 floats) in one ping-pong buffer
  static constexpr bool obs_ok = TH * TW * 3 <= ROWS * P && ONRY * TW * 3 <= ROWS * P;
  static constexpr size_t obs_lds_bytes() { return (size_t)(ONRY * OKY + ONCX * OKX + ONRY + ONCX + 1) / 4 * 16 + 16; }
  static constexpr size_t mask_lds_bytes() { return kTailCacheJ ? (size_t)RH * P : (((size_t)RH * (RW / 4) + 15) / 16) * 16; }
  static constexpr size_t lds_bytes() {
    return (size_t)(2 * ROWS * P + (kTailCacheJ ? RH * P : 0)) * sizeof(float) + mask_lds_bytes() + obs_lds_bytes();
  }
};

template <int... KS>
__global__ __launch_bounds__(kTailThreads, kTailWavesPerSimd) void taxim_tail_kernel(TailArgs a) {
  using C = TailCfg<KS...>;
  constexpr int NL = C::NL, TW = C::TW, TH = C::TH, HLY = C::HLY, HLX = C::HLX, RW = C::RW, RH = C::RH;
  constexpr int PADX = C::PADX, PADY = C::PADY, P = C::P, ROWS = C::ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* bufA = reinterpret_cast<float*>(smem_raw);
  float* bufB = bufA + ROWS * P;
  float* bufJ = bufB + ROWS * P;                       // RH x P (kTailCacheJ only)
  uint8_t* bufM = reinterpret_cast<uint8_t*>(bufJ + (kTailCacheJ ? RH * P : 0));  // RH x P bytes, or RH x RW/4 nibbles
  constexpr int MG = RW / 4;                           // mask groups per region row (!kTailCacheJ)

  const int H = a.H, W = a.W;
  const int ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int frame = lid / (ntx * nty);
  const int tix = lid - frame * ntx * nty;
  const int ty0 = (tix / ntx) * TH, tx0 = (tix % ntx) * TW;
  const size_t fo = (size_t)frame * H * W;
  const float sa = a.shift_a[frame], sb = a.shift_b[frame];
  const float thr = -a.pdepth[frame] * a.contact_scale;  // TT:459
  const float* __restrict__ zin = a.zin + fo;
  const float* __restrict__ hm = a.hm + fo;
  const int gy0 = ty0 - HLY, gx0 = tx0 - HLX;  // global coords of region cell (0,0)
  const int tid = threadIdx.x;
  constexpr int NT = kTailThreads;

  // ---- policy observation, step 0: this tile's tap windows (see the reduction after the epilogue).  A window is KY (KX)
  // consecutive tile rows (columns) covering support(cell) ^ tile, so the reductions run fixed, fully unrolled trip
  // counts with immediate LDS offsets.  Built here, in its own LDS area, so that the dependent table loads overlap the
  // tile load instead of sitting exposed between the epilogue and the reduction.
  float* wly = reinterpret_cast<float*>(bufM + C::mask_lds_bytes());  // [NRY][KY] taps of cell row j over the KY tile rows from wby[j]
  float* wlx = wly + C::ONRY * C::OKY;                    // [NCX][KX] same for the cell columns
  int* wby = reinterpret_cast<int*>(wlx + C::ONCX * C::OKX);  // [NRY] first row of the window (tile-relative)
  int* wbx = wby + C::ONRY;                                   // [NCX]
  // ---- load: previous level, J and M, ALL at REFLECTED coordinates ----
  // Out-of-image halo cells hold the mirror image of the in-image data (= torch 'reflect' padding, TT:411).  A
  // symmetric kernel maps a mirror-symmetric signal to a mirror-symmetric signal, and the restore Z[M] = J[M] uses
  // the mirrored J / M, so the halo stays the reflect padding of every later level without any re-mirroring pass
  // (the only difference to padding each level explicitly is the summation order of the taps: float roundoff).
  // One 4-cell group per thread and pass; all passes unrolled so every global load of the tile is in flight at once.
  {
    constexpr int G = RW / 4;          // 16-byte groups per region row
    constexpr int RPP = NT / G;        // region rows covered per pass
    constexpr int NPASS = (RH + RPP - 1) / RPP;
    const int lrow = tid / G, lx = (tid - lrow * G) * 4;
    const int gx = gx0 + lx;
    const bool xin = gx >= 0 && gx + 3 < W;  // whole group inside the image (W % 4 == 0, gx0 % 4 == 0)
    unsigned rxo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rxo[k] = (unsigned)min(max(reflect_idx(gx + k, W), 0), W - 1);
    v4f zv[NPASS], hv[NPASS], gv[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int ly = ps * RPP + lrow;
      if (lrow < RPP && ly < RH) {
        const unsigned ro = (unsigned)min(max(reflect_idx(gy0 + ly, H), 0), H - 1) * (unsigned)W;
        if (xin) {
          zv[ps] = *reinterpret_cast<const v4f*>(zin + ro + gx);
          hv[ps] = *reinterpret_cast<const v4f*>(hm + ro + gx);
          gv[ps] = a.gel ? *reinterpret_cast<const v4f*>(a.gel + ro + gx) : (v4f)(0.0f);  // nullptr: gel == 0
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            zv[ps][k] = zin[ro + rxo[k]];
            hv[ps][k] = hm[ro + rxo[k]];
            gv[ps][k] = a.gel ? a.gel[ro + rxo[k]] : 0.0f;
          }
        }
      }
    }
    // (observation tap windows: issued behind the tile loads, in front of their first use)
    if (a.obs_part && a.sh.rgb) {
      constexpr int NRY = C::ONRY, NCX = C::ONCX, KY = C::OKY, KX = C::OKX;
      const ObsTables& T = a.obs;
      const float scy = (float)H / (float)T.oh, scx = (float)W / (float)T.ow;
      const int oy0 = max(0, (int)floorf(((float)ty0 - scy) / scy));
      const int ox0 = max(0, (int)floorf(((float)tx0 - scx) / scx));
      if (tid < NRY * KY) {
        const int j = tid / KY, t = tid - j * KY, oy = oy0 + j;
        float w = 0.0f;
        int base = 0;
        if (oy < T.oh) {
          const int flo = T.ylo[oy];
          base = min(max(flo, ty0), ty0 + TH - KY);
          const int k = base + t - flo;  // tap index of tile row base + t
          if (k >= 0 && k < T.ycnt[oy] && base + t < H) w = T.wy[(size_t)oy * T.ky + k];
          base -= ty0;
        }
        wly[tid] = w;
        if (t == 0) wby[j] = base;
      }
      if (tid < NCX * KX) {
        const int q = tid / KX, t = tid - q * KX, ox = ox0 + q;
        float w = 0.0f;
        int base = 0;
        if (ox < T.ow) {
          const int flo = T.xlo[ox];
          base = min(max(flo, tx0), tx0 + TW - KX);
          const int k = base + t - flo;
          if (k >= 0 && k < T.xcnt[ox] && base + t < W) w = T.wx[(size_t)ox * T.kx + k];
          base -= tx0;
        }
        wlx[tid] = w;
        if (t == 0) wbx[q] = base;
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int ly = ps * RPP + lrow;
      if (lrow < RPP && ly < RH) {
        *reinterpret_cast<v4f*>(bufA + (ly + PADY) * P + PADX + lx) = zv[ps];
        v4f Jv;
        uint8_t mk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float S = (hv[ps][k] - sa) - sb;
          const float J = fmin_raw(S, gv[ps][k]);
          Jv[k] = J;
          mk[k] = (((J - gv[ps][k]) < thr) && (S < 0.0f)) ? 1 : 0;  // TT:457-461
        }
        if constexpr (kTailCacheJ) {
          *reinterpret_cast<v4f*>(bufJ + ly * P + PADX + lx) = Jv;
          *reinterpret_cast<uchar4*>(bufM + ly * P + PADX + lx) = (uchar4){mk[0], mk[1], mk[2], mk[3]};
        } else {
          bufM[ly * MG + (lx >> 2)] = (uint8_t)(mk[0] | (mk[1] << 1) | (mk[2] << 2) | (mk[3] << 3));
        }
      }
    }
  }
  __syncthreads();

  static_for<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int K = C::K[l];
    constexpr int R = (K - 1) / 2;
    constexpr int R4 = (R + 3) & ~3;
    // margin (beyond the output tile) that must still be valid AFTER this level: radii of the later levels + 2
    constexpr int MY = C::margin_after(l), MX = MY + (HLX - HLY);
    // V-pass output rows [VY0, VY1), H-pass rows [HY0, HY1) (= V rows +- R), columns [X0, X1) - 4-aligned supersets
    constexpr int VY0 = (HLY - MY) & ~3, VY1 = (HLY + TH + MY + 3) & ~3;
    constexpr int HY0 = (HLY - MY - R) < 0 ? 0 : (HLY - MY - R), HY1 = (HLY + TH + MY + R) > RH ? RH : (HLY + TH + MY + R);
    constexpr int X0 = (HLX - MX) & ~3, X1 = (HLX + TW + MX + 3) & ~3;
    constexpr int NXG = (X1 - X0) / 4;
    const float* __restrict__ taps = a.taps[l];
    float* const cur = C::in_buf(l) ? bufB : bufA;   // holds this level's input
    float* const oth = C::in_buf(l) ? bufA : bufB;
    auto restore_store = [&](v4f o, int row, int x0, float* dst) {
      if constexpr (l < NL - 1) {  // TT:467 Z[M] = J[M]; the final blur (TT:468-471) has no restore
        if constexpr (kTailCacheJ) {
          const int ci = row * P + PADX + x0;
          const v4f Jv = *reinterpret_cast<const v4f*>(bufJ + ci);
          const uchar4 Mv = *reinterpret_cast<const uchar4*>(bufM + ci);
          o.x = Mv.x ? Jv.x : o.x; o.y = Mv.y ? Jv.y : o.y; o.z = Mv.z ? Jv.z : o.z; o.w = Mv.w ? Jv.w : o.w;
        } else {
          const unsigned bits = bufM[row * MG + (x0 >> 2)];
          if (bits) {  // contact pixels only: J = min(S, gel) again, from the (reflected) height map / gel in L2
            const unsigned ro = (unsigned)min(max(reflect_idx(gy0 + row, H), 0), H - 1) * (unsigned)W;
            const int gx = gx0 + x0;
            v4f hq, gq;
            if (gx >= 0 && gx + 3 < W) {
              hq = *reinterpret_cast<const v4f*>(hm + ro + gx);
              gq = a.gel ? *reinterpret_cast<const v4f*>(a.gel + ro + gx) : (v4f)(0.0f);
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const unsigned xo = (unsigned)min(max(reflect_idx(gx + k, W), 0), W - 1);
                hq[k] = hm[ro + xo];
                gq[k] = a.gel ? a.gel[ro + xo] : 0.0f;
              }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (bits & (1u << k)) o[k] = fmin_raw((hq[k] - sa) - sb, gq[k]);
          }
        }
      }
      *reinterpret_cast<v4f*>(dst + (row + PADY) * P + PADX + x0) = o;
    };
    // horizontal sums of 4 consecutive x of one row of `src` (window read as 16-byte groups)
    auto hsum4 = [&](const float* src, int ly, int x0) -> v4f {
      const float* row = src + (ly + PADY) * P + PADX + x0 - R4;
      float win[4 + 2 * R4];
      static_for<0, (4 + 2 * R4) / 4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const v4f q = *reinterpret_cast<const v4f*>(row + 4 * j);
        win[4 * j] = q.x; win[4 * j + 1] = q.y; win[4 * j + 2] = q.z; win[4 * j + 3] = q.w;
      });
      v4f o = (v4f)(0.0f);
      static_for<0, K>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const float w = taps[t < K - 1 - t ? t : K - 1 - t];
        o.x += w * win[R4 - R + t];
        o.y += w * win[R4 - R + t + 1];
        o.z += w * win[R4 - R + t + 2];
        o.w += w * win[R4 - R + t + 3];
      });
      return o;
    };
    if constexpr (C::fused2d(l)) {
      // ---- single phase: cur -> oth, 4 columns x 2 rows per item; the 2 + 2R horizontal row sums never leave registers ----
      for (int it = tid; it < ((VY1 - VY0) / 2) * NXG; it += NT) {
        const int ly0 = VY0 + (it / NXG) * 2, x0 = X0 + (it % NXG) * 4;
        v4f acc[2] = {(v4f)(0.0f), (v4f)(0.0f)};
        static_for<0, 2 + 2 * R>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const v4f q = hsum4(cur, ly0 - R + j, x0);
          static_for<0, 2>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int t = j - r;
            if constexpr (t >= 0 && t < K) acc[r] += taps[t < K - 1 - t ? t : K - 1 - t] * q;
          });
        });
        restore_store(acc[0], ly0, x0, oth);
        restore_store(acc[1], ly0 + 1, x0, oth);
      }
      __syncthreads();
    } else if constexpr (K > 1) {
      // ---- H-pass: cur -> oth, 4 consecutive x per item ----
      for (int it = tid; it < (HY1 - HY0) * NXG; it += NT) {
        const int ly = HY0 + it / NXG, x0 = X0 + (it % NXG) * 4;
        *reinterpret_cast<v4f*>(oth + (ly + PADY) * P + PADX + x0) = hsum4(cur, ly, x0);
      }
      __syncthreads();
      // ---- V-pass: oth -> cur, 4 columns x 2 rows per item, + masked restore ----
      // (2-row items: ~2x the items of a 4x4 blocking, so every wave carries one item instead of half of them carrying
      // 16 outputs each while the rest wait at the barrier; the extra window reads are LDS-cheap)
      for (int it = tid; it < ((VY1 - VY0) / 2) * NXG; it += NT) {
        const int ly0 = VY0 + (it / NXG) * 2, x0 = X0 + (it % NXG) * 4;
        v4f acc[2] = {(v4f)(0.0f), (v4f)(0.0f)};
        static_for<0, 2 + 2 * R>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const v4f q = *reinterpret_cast<const v4f*>(oth + (ly0 - R + j + PADY) * P + PADX + x0);
          static_for<0, 2>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int t = j - r;
            if constexpr (t >= 0 && t < K) acc[r] += taps[t < K - 1 - t ? t : K - 1 - t] * q;
          });
        });
        restore_store(acc[0], ly0, x0, cur);
        restore_store(acc[1], ly0 + 1, x0, cur);
      }
      __syncthreads();
    } else if constexpr (l < NL - 1) {
      // K == 1: the blur is the identity (single tap = 1.0) but the restore still runs
      static_assert(kTailCacheJ, "the k == 1 restore reads the cached J");
      for (int c = tid; c < RH * RW; c += NT) {
        const int ly = c / RW, lx = c - ly * RW;
        if (bufM[ly * P + PADX + lx]) cur[(ly + PADY) * P + PADX + lx] = bufJ[ly * P + PADX + lx];
      }
      __syncthreads();
    }
  });

  // ---- epilogue: 4 consecutive pixels per thread: float4 I/O for deformed gel / mask / background / RGB ----
  float* const bufZ = C::in_buf(NL) ? bufB : bufA;  // final deformed gel of the region
  float* const bufS = C::in_buf(NL) ? bufA : bufB;  // the other ping-pong buffer: free, stages RGB for the observation
  float f_zmax = -INFINITY;  // FOTS contact statistics of this thread's strips (fots_part)
  int f_cnt = 0, f_sr = 0, f_sc = 0;
  const bool fast_w = (W % 4) == 0;
  for (int sidx = tid; sidx < (TH * TW) / 4; sidx += NT) {  // strip index within the tile
    const int oy = sidx / (TW / 4), ox = (sidx - oy * (TW / 4)) * 4;
    const int gy = ty0 + oy, gx = tx0 + ox;
    if (gy >= H || gx >= W) {  // tile overhang: only the observation staging must not hold stale (possibly non-finite) data
      if (a.obs_part && a.sh.rgb) {
        float* sg = bufS + oy * (TW * 3) + ox * 3;
        reinterpret_cast<v4f*>(sg)[0] = (v4f)(0.0f); reinterpret_cast<v4f*>(sg)[1] = (v4f)(0.0f); reinterpret_cast<v4f*>(sg)[2] = (v4f)(0.0f);
      }
      continue;
    }
    const int ly = oy + HLY, lx = ox + HLX;
    const unsigned p = (unsigned)gy * (unsigned)W + (unsigned)gx;  // 32-bit pixel offset from the wave-uniform frame bases
    const float* crow = bufZ + (ly + PADY) * P + PADX + lx;
    if (a.fots_part) {
      uint8_t mb[4];
      if constexpr (kTailCacheJ) {
        const uchar4 mv = *reinterpret_cast<const uchar4*>(bufM + ly * P + PADX + lx);
        mb[0] = mv.x; mb[1] = mv.y; mb[2] = mv.z; mb[3] = mv.w;
      } else {
        const unsigned bits = bufM[ly * MG + (lx >> 2)];
        mb[0] = bits & 1; mb[1] = (bits >> 1) & 1; mb[2] = (bits >> 2) & 1; mb[3] = (bits >> 3) & 1;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (gx + i < W) {
          f_zmax = fmaxf(f_zmax, crow[i]);
          const int m1 = mb[i] != 0;
          f_cnt += m1; f_sr += m1 * gy; f_sc += m1 * (gx + i);
        }
      }
    }
    if (fast_w) {  // whole strip inside the image
      if (a.z_out) *reinterpret_cast<v4f*>(reinterpret_cast<char*>(a.z_out + fo) + p * 4u) = *reinterpret_cast<const v4f*>(crow);
      if (a.mask_out) {
        uchar4 mv;
        if constexpr (kTailCacheJ) mv = *reinterpret_cast<const uchar4*>(bufM + ly * P + PADX + lx);
        else {
          const unsigned bits = bufM[ly * MG + (lx >> 2)];
          mv = (uchar4){(uint8_t)(bits & 1), (uint8_t)((bits >> 1) & 1), (uint8_t)((bits >> 2) & 1), (uint8_t)((bits >> 3) & 1)};
        }
        *reinterpret_cast<uchar4*>(a.mask_out + fo + p) = mv;
      }
    } else {
      for (int i = 0; i < 4 && gx + i < W; ++i) {
        if (a.z_out) a.z_out[fo + p + i] = crow[i];
        if (a.mask_out) a.mask_out[fo + p + i] = kTailCacheJ ? bufM[ly * P + PADX + lx + i] : (uint8_t)((bufM[ly * MG + (lx >> 2)] >> i) & 1);
      }
    }
    if (a.sh.rgb) {
      // replicate padding of the gradient maps == evaluate at the clamped pixel (TT:501-502)
      const int yc = min(max(gy, 1), H - 2) - gy0;
      const float* rc = bufZ + (yc + PADY) * P + PADX;
      float rgb[12];
      if (fast_w) {
        float zn[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int xc = min(max(gx + i, 1), W - 2) - gx0;
          zn[i][0] = rc[xc - P]; zn[i][1] = rc[xc + P]; zn[i][2] = rc[xc - 1]; zn[i][3] = rc[xc + 1];
        }
        shade_strip4_rgb(a.sh, zn, gx, gy, rgb);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gxi = min(gx + i, W - 1);
          const int xc = min(max(gxi, 1), W - 2) - gx0;
          shade_pixel_rgb(a.sh, rc[xc - P], rc[xc + P], rc[xc - 1], rc[xc + 1], gxi, gy, rgb + 3 * i);
        }
      }
      if (a.obs_part) {  // stage the strip for the observation reduction below (the other buffer is free after the last level)
        float* sg = bufS + oy * (TW * 3) + ox * 3;
        reinterpret_cast<v4f*>(sg)[0] = (v4f){rgb[0], rgb[1], rgb[2], rgb[3]};
        reinterpret_cast<v4f*>(sg)[1] = (v4f){rgb[4], rgb[5], rgb[6], rgb[7]};
        reinterpret_cast<v4f*>(sg)[2] = (v4f){rgb[8], rgb[9], rgb[10], rgb[11]};
      }
      float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(a.sh.rgb + fo * 3) + p * 12u);
      if (fast_w) {
        reinterpret_cast<v4f*>(o)[0] = (v4f){rgb[0], rgb[1], rgb[2], rgb[3]};
        reinterpret_cast<v4f*>(o)[1] = (v4f){rgb[4], rgb[5], rgb[6], rgb[7]};
        reinterpret_cast<v4f*>(o)[2] = (v4f){rgb[8], rgb[9], rgb[10], rgb[11]};
      } else {
        for (int i = 0; i < 4 && gx + i < W; ++i) { o[3 * i] = rgb[3 * i]; o[3 * i + 1] = rgb[3 * i + 1]; o[3 * i + 2] = rgb[3 * i + 2]; }
      }
    }
  }
  if (a.pix_z) {  // the few marker pixels of this tile: straight from the LDS copies of the final level and the mask
    const int cnt = a.mk_cnt[tix];
    if (tid < cnt) {
      const int e = a.mk_tile[tix * kTailMaxMarkersPerTile + tid];
      const int m = e >> 16, ly = ((e >> 8) & 0xff) + HLY, lx = (e & 0xff) + HLX;
      uint8_t mv;
      if constexpr (kTailCacheJ) mv = bufM[ly * P + PADX + lx];
      else mv = (uint8_t)((bufM[ly * MG + (lx >> 2)] >> (lx & 3)) & 1);
      a.pix_z[(size_t)frame * a.n_markers + m] = bufZ[(ly + PADY) * P + PADX + lx];
      a.pix_m[(size_t)frame * a.n_markers + m] = mv;
    }
  }
  if (a.fots_part) {  // one record per wave: no barrier, no atomics; fots_combine_kernel adds the records of an env
    f_zmax = wave_scan_max_lane63(f_zmax);
    f_cnt = wave_scan_add_lane63(f_cnt);
    f_sr = wave_scan_add_lane63(f_sr);
    f_sc = wave_scan_add_lane63(f_sc);
    if ((tid & 63) == 63) {
      FotsReduce r;
      r.zmax = f_zmax; r.count = f_cnt; r.sum_row = f_sr; r.sum_col = f_sc;
      a.fots_part[(size_t)lid * kTailWavesPerTile + (tid >> 6)] = r;
    }
  }
  // ---- policy observation (torchvision antialiased bilinear = separable triangle filter of support H/oh x W/ow):
  //      the tile reduces its own pixels vertically, then horizontally, to the <= NRY x NCX observation cells it
  //      overlaps and stores these PARTIAL sums (un-normalised) in its own slot of obs_part; obs_finish_kernel adds
  //      the <= 4 partials of a cell in a fixed order.  No atomics, no memset, the full-resolution frame is not re-read.
  if constexpr (C::obs_ok) if (a.obs_part && a.sh.rgb) {
    constexpr int NRY = C::ONRY, NCX = C::ONCX, KY = C::OKY, KX = C::OKX, TWC = TW * 3;
    static_assert(TWC % 64 == 0 && NT % 64 == 0, "a wave stays inside one observation row in the vertical pass");
    static_assert(KY <= TH && KX <= TW && KY % 4 == 0 && KX % 4 == 0, "tap windows");
    static_assert(kTailThreads == 64 * kTailWavesPerTile, "FOTS partial records per tile");
    float* v1 = bufZ;                  // [NRY][TWC]; the final level is dead once every strip of the epilogue is shaded
    __syncthreads();
    // vertical: item = (cell row j, 4 consecutive column*channel values): 16-byte LDS reads, one item per thread
    static_assert(TWC % 4 == 0 && NRY * (TWC / 4) <= NT, "vertical observation items");
    if (tid < NRY * (TWC / 4)) {
      const int j = tid / (TWC / 4), xc = (tid - j * (TWC / 4)) * 4;
      const float* sg = bufS + wby[j] * TWC + xc;
      const v4f* wv = reinterpret_cast<const v4f*>(wly + j * KY);
      v4f acc = (v4f)(0.0f);
      static_for<0, KY / 4>([&](auto tc) {
        constexpr int t = decltype(tc)::value * 4;
        const v4f w = wv[t / 4];
        acc += w.x * *reinterpret_cast<const v4f*>(sg + t * TWC);
        acc += w.y * *reinterpret_cast<const v4f*>(sg + (t + 1) * TWC);
        acc += w.z * *reinterpret_cast<const v4f*>(sg + (t + 2) * TWC);
        acc += w.w * *reinterpret_cast<const v4f*>(sg + (t + 3) * TWC);
      });
      *reinterpret_cast<v4f*>(v1 + j * TWC + xc) = acc;
    }
    __syncthreads();
    float* part = a.obs_part + (size_t)lid * (NRY * NCX * 3);  // lid = frame * tiles_per_frame + tile index
    if (tid < NRY * NCX * 3) {  // horizontal: item = (cell row j, cell column q, channel)
      const int ch = tid % 3, q = (tid / 3) % NCX, j = tid / (3 * NCX);
      const float* sg = v1 + j * TWC + wbx[q] * 3 + ch;
      const v4f* wv = reinterpret_cast<const v4f*>(wlx + q * KX);
      float acc = 0.0f;
      static_for<0, KX / 4>([&](auto tc) {
        constexpr int t = decltype(tc)::value * 4;
        const v4f w = wv[t / 4];
        acc = fmaf(w.x, sg[t * 3], acc);
        acc = fmaf(w.y, sg[(t + 1) * 3], acc);
        acc = fmaf(w.z, sg[(t + 2) * 3], acc);
        acc = fmaf(w.w, sg[(t + 3) * 3], acc);
      });
      part[tid] = acc;
    }
  }
}

// adds the per-tile partial sums of every observation cell in a fixed order and normalises by the weight sums.
// One thread per (frame, cell) = 3 channels; 32-bit index arithmetic only (a 64-bit div/mod chain per element made the
// first version 12 us for 786 K outputs).
template <bool U8>
__global__ __launch_bounds__(256) void obs_finish_kernel(const float* __restrict__ part, void* __restrict__ obs_v, ObsTables T,
                                                        int H, int W, int ntx, int nty, int TW, int TH, int NRY, int NCX) {
  const int oh = T.oh, ow = T.ow;
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= oh * ow) return;
  const int b = blockIdx.y;
  const int oy = cell / ow, ox = cell - oy * ow;
  const float scy = (float)H / oh, scx = (float)W / ow;
  const int ylo = T.ylo[oy], yhi = ylo + T.ycnt[oy], xlo = T.xlo[ox], xhi = xlo + T.xcnt[ox];
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  for (int ty = ylo / TH; ty <= (yhi - 1) / TH; ++ty) {
    const int oy0 = max(0, (int)floorf(((float)(ty * TH) - scy) / scy));
    for (int tx = xlo / TW; tx <= (xhi - 1) / TW; ++tx) {
      const int ox0 = max(0, (int)floorf(((float)(tx * TW) - scx) / scx));
      const int j = oy - oy0, q = ox - ox0;
      if (j < 0 || j >= NRY || q < 0 || q >= NCX) continue;  // cannot happen for the scales run_tail accepts
      const size_t tile = ((size_t)b * nty + ty) * ntx + tx;
      const float* pp = part + (tile * (NRY * NCX) + (unsigned)(j * NCX + q)) * 3;
      a0 += pp[0]; a1 += pp[1]; a2 += pp[2];
    }
  }
  const float nrm = T.xsum[ox] * T.ysum[oy];
  const size_t oi = ((size_t)b * (oh * ow) + cell) * 3;
  if constexpr (U8) {  // RGB is clipped to [0,1] (TT:257-258), so is every convex combination of it
    uint8_t* o = static_cast<uint8_t*>(obs_v) + oi;
    o[0] = (uint8_t)(a0 / nrm * 255.0f + 0.5f); o[1] = (uint8_t)(a1 / nrm * 255.0f + 0.5f); o[2] = (uint8_t)(a2 / nrm * 255.0f + 0.5f);
  } else {
    float* o = static_cast<float*>(obs_v) + oi;
    o[0] = a0 / nrm; o[1] = a1 / nrm; o[2] = a2 / nrm;
  }
}

__global__ __launch_bounds__(256) void obs_to_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (uint8_t)(fminf(fmaxf(src[i], 0.0f), 1.0f) * 255.0f + 0.5f);
}

hipError_t run_obs_to_u8(const float* src, uint8_t* dst, size_t n, hipStream_t st) {
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(obs_to_u8_kernel, dim3(grid), dim3(256), 0, st, src, dst, n);
  return hipGetLastError();
}

// observation geometry of the fused tail instantiation for (n_fused, first fused k); false: no room to stage the observation
bool tail_obs_geom(int n_fused, int k0, int* nry, int* ncx, int* ky, int* kx) {
  auto set = [&](auto cfg) {
    using C = decltype(cfg);
    *nry = C::ONRY; *ncx = C::ONCX; *ky = C::OKY; *kx = C::OKX;
    return C::obs_ok;
  };
  if (n_fused == 4 && k0 == 9) return set(TailCfg<9, 5, 3, 5>{});
  if (n_fused == 4 && k0 == 15) return set(TailCfg<15, 9, 5, 9>{});
  if (n_fused == 3 && k0 == 9) return set(TailCfg<9, 5, 9>{});
  if (n_fused == 3 && k0 == 5) return set(TailCfg<5, 3, 5>{});
  return false;
}

hipError_t run_obs_finish(const float* part, void* obs, bool u8, const ObsTables& t, int H, int W, int B, int nry, int ncx,
                          hipStream_t st) {
  const int TW = 64, TH = 32;
  const int ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH;
  const dim3 grid((t.oh * t.ow + 255) / 256, B);
  if (u8) hipLaunchKernelGGL(obs_finish_kernel<true>, grid, dim3(256), 0, st, part, obs, t, H, W, ntx, nty, TW, TH, nry, ncx);
  else hipLaunchKernelGGL(obs_finish_kernel<false>, grid, dim3(256), 0, st, part, obs, t, H, W, ntx, nty, TW, TH, nry, ncx);
  return hipGetLastError();
}

// can the fused tail (n_fused levels, first fused k) produce this observation?  Cells per tile and filter lengths must fit
// what the instantiation is compiled for.
bool obs_fusable(const ObsTables& t, int H, int W, int n_fused, int k0) {
  int nry, ncx, ky, kx;
  if (!tail_obs_geom(n_fused, k0, &nry, &ncx, &ky, &kx) || W % 4 != 0) return false;
  const float scy = (float)H / t.oh, scx = (float)W / t.ow;
  return 32.0f / scy + 3.0f <= (float)nry && 64.0f / scx + 3.0f <= (float)ncx && t.ymax <= ky && t.xmax <= kx;
}

size_t tail_tiles_per_frame(int H, int W) { return (size_t)((W + 63) / 64) * ((H + 31) / 32); }

size_t obs_part_floats(int H, int W, int B, int nry, int ncx) {
  return (size_t)B * tail_tiles_per_frame(H, W) * nry * ncx * 3;
}

template <int... KS>
static hipError_t launch_tail(const TailArgs& a0, hipStream_t st) {
  using C = TailCfg<KS...>;
  if (!C::obs_ok && a0.obs_part) return hipErrorInvalidValue;  // callers check obs_fusable() first
  TailArgs a = a0;
  const int ntx = (a.W + C::TW - 1) / C::TW, nty = (a.H + C::TH - 1) / C::TH;
  auto kern = taxim_tail_kernel<KS...>;
  // > 64 KB of dynamic LDS is an opt-in per kernel AND device: one flag per device (contexts on several GPUs in one process)
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)C::lds_bytes());
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(ntx * nty * a.B), dim3(kTailThreads), C::lds_bytes(), st, a);
  return hipGetLastError();
}

// number of trailing levels the fused tail covers for this context (0 = no fused instantiation)
int tail_levels(const LevelDesc* lv, int n_levels, int H, int W) {
  if (H < 8 || W < 8 || W % 4 != 0) return 0;
  auto match = [&](int n, const int* ks) {
    if (n_levels < n) return false;
    for (int i = 0; i < n; ++i) {
      const LevelDesc& d = lv[n_levels - n + i];
      if (!d.same_taps || d.kw != ks[i]) return false;
    }
    return true;
  };
  static const int k320[4] = {9, 5, 3, 5};    // 320x240 (gsmini presets, BASELINE configs 1-4)
  static const int k640[4] = {15, 9, 5, 9};   // 640x480 (BASELINE config 5)
  // 640x480: fusing all four levels needs a 104 x 72 region (halo 19) = 130 KB of LDS, ONE workgroup per CU and 3.7x the
  // pixels of the tile at the first level; leaving k = 15 to a band kernel gives the 320x240 footprint (88 x 56 region,
  // two workgroups per CU).  TACEX_TAIL_LEVELS_640=4 restores the four-level kernel (A/B hook).
  static const int lv640 = getenv("TACEX_TAIL_LEVELS_640") ? atoi(getenv("TACEX_TAIL_LEVELS_640")) : 3;
  static const int lv320 = getenv("TACEX_TAIL_LEVELS_320") ? atoi(getenv("TACEX_TAIL_LEVELS_320")) : 4;
  if (match(4, k320)) return lv320 == 3 ? 3 : 4;
  if (match(4, k640)) return lv640 == 4 ? 4 : 3;
  return 0;
}

hipError_t run_tail(const LevelDesc* lv, int n_levels, int n_fused, const float* zin, const float* hm, const float* gel,
                    const float* sa, const float* sb, const float* pd, float* z_out, uint8_t* mask_out,
                    const ShadeParams* sp, float* rgb, float* obs_part, const ObsTables* obs_tab, FotsReduce* fots_part,
                    const FotsTaps* taps, float* pix_z, uint8_t* pix_m, int B, int H, int W, float contact_scale, hipStream_t st) {
  TailArgs a{};
  a.fots_part = fots_part;
  if (taps && taps->mk_tile && pix_z && pix_m) {
    a.pix_z = pix_z; a.pix_m = pix_m; a.mk_tile = taps->mk_tile; a.mk_cnt = taps->mk_cnt; a.n_markers = taps->n_markers;
  }
  a.obs_part = obs_tab ? obs_part : nullptr;
  if (a.obs_part) a.obs = *obs_tab;
  a.zin = zin; a.hm = hm; a.gel = lv[0].gel_zero ? nullptr : gel; a.shift_a = sa; a.shift_b = sb; a.pdepth = pd; a.z_out = z_out; a.mask_out = mask_out;
  a.H = H; a.W = W; a.B = B; a.contact_scale = contact_scale;
  for (int i = 0; i < n_fused; ++i) a.taps[i] = lv[n_levels - n_fused + i].taps_w_dev;
  if (sp && rgb) {
    a.sh.poly = sp->poly_dev; a.sh.bg = sp->bg_nhwc_dev; a.sh.fx = sp->fx_dev; a.sh.fy = sp->fy_dev; a.sh.rgb = rgb;
    a.sh.idx_out = nullptr; a.sh.H = H; a.sh.W = W; a.sh.B = B; a.sh.nb = sp->nb; a.sh.pixmm = sp->pixmm;
    a.sh.calib_h = (float)sp->calib_h; a.sh.calib_w = (float)sp->calib_w; a.sh.x_binr = sp->x_binr; a.sh.y_binr = sp->y_binr;
    a.sh.gsy = (float)(0.5 * H / sp->calib_h / (double)sp->pixmm); a.sh.gsx = (float)(0.5 * W / sp->calib_w / (double)sp->pixmm);
    a.sh.inv_x_binr = (float)(1.0 / (double)sp->x_binr); a.sh.inv_y_binr = (float)(1.0 / (double)sp->y_binr);
  }
  const int k0 = lv[n_levels - n_fused].kw;
  if (n_fused == 4 && k0 == 9) return launch_tail<9, 5, 3, 5>(a, st);
  if (n_fused == 4 && k0 == 15) return launch_tail<15, 9, 5, 9>(a, st);
  if (n_fused == 3 && k0 == 9) return launch_tail<9, 5, 9>(a, st);
  if (n_fused == 3 && k0 == 5) return launch_tail<5, 3, 5>(a, st);
  return hipErrorInvalidValue;
}

}  // namespace tacex
