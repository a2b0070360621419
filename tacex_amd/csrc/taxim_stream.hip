// Streaming fused tail of the Taxim path: the small-kernel pyramid levels (k = 9, 5, 3 + final 5 at 320x240) with their masked
// restores, the shading, the policy observation and the FOTS by-products - as a WAVE-AUTONOMOUS line-buffer pipeline.
//
// Why (round-2 counters of the LDS-tiled tail, taxim_tail.hip): 52 % of its wave time sat in barriers / waitcnt, 40 % of its
// LDS cycles were bank conflicts, and a 64x32 tile drags a 2.4x halo through every phase.  Here nothing is shared between waves:
//   * one wave owns a vertical STRIP of the frame, 3 pixels per lane (62 lanes = 160 output columns + a 12-pixel halo on both
//     sides) and marches down the rows; there is no __syncthreads and no LDS tile;
//   * the horizontal pass of a level reads its neighbours from the adjacent LANES (DPP wave shifts, no LDS);
//   * the vertical pass keeps K - 1 partial sums per pixel in registers and SCATTERS each new row into them
//     (out = fma(w[K-1], h, A[K-2]); A[j] = fma(w[j], h, A[j-1]) ...): the window "shifts" through the FMA destinations, so a row
//     costs exactly K FMAs per pixel and no register moves;
//   * the masked restore Z[M] = J[M] (TT:467) needs S = hm - min - press of a row 4..7 iterations after it was loaded: an
//     8-row ring per wave in LDS (wave-private: no barrier, no conflicts - every lane touches only its own 16 bytes);
//   * level l consumes level l-1's output row of the SAME iteration, so a row entering at iteration y leaves the last level
//     as row y - sum(R) and is shaded as row y - sum(R) - 1 (central differences need the row below).
// HBM sees one read of the previous level + the height map and one write of RGB, as with the tiled tail; the halo is only
// horizontal (2 x 12 of 184 columns, those re-reads are L2 hits) plus sum(R) + 1 warm-up rows per vertical segment.
//
// Summation order of every level equals the band kernels / the tiled tail (taps ascending, fmaf chains), so the deformed gel
// is bit-identical to theirs.  Reference semantics: TT:464-471 (levels + restore), TT:411 (reflect padding: the strip loads
// rows / columns at REFLECTED coordinates, a symmetric kernel keeps the halo mirror-symmetric), TT:475-503, 237-258 (shading).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "tacex_internal.h"
#include "taxim_device.h"

namespace tacex {

constexpr int kStreamPx = 3;         // pixels per lane
constexpr int kStreamRing = 8;       // rows of S kept per wave (deepest restore looks back sum(R) - R_last rows <= 7)
constexpr int kStreamWaves = 4;      // waves per workgroup (independent of each other)
constexpr int kStreamObsActive = 3;  // observation rows a frame row can contribute to (down-sampling factor >= 2)

struct StreamArgs {
  const float* zin;      // (B,H,W) output of the last band level
  const float* hm;       // (B,H,W)
  const float* gel;      // (H,W); nullptr = identically zero
  const float* shift_a;  // (B,)
  const float* shift_b;
  const float* pdepth;
  const float* taps[kStreamMaxLevels];
  ShadeArgs sh;
  int H, W, B;
  float contact_scale;
  int nstrips, strip_w;  // strips per frame, valid columns per strip
  int nseg, seg_rows;    // vertical segments per strip, rows per segment
  // policy observation (nullable): per (frame, strip, segment) block of partial sums [obs_nrows][obs_ncols][3]
  float* obs_part;
  const int* obs_row_o0;    // (H,)   first observation row a frame row contributes to
  const float* obs_row_w;   // (H,3)  its weights for rows o0, o0 + 1, o0 + 2 (0 where out of the filter support)
  const int* obs_xlo; const int* obs_xcnt; const float* obs_wx; int obs_kx;   // column filters (ObsTables)
  const int* obs_strip_q0;  // (nstrips,) first observation column a strip's valid columns touch
  const int* obs_strip_nq;  // (nstrips,) number of such columns
  const int* obs_seg_oa;    // (nseg,) first / last observation row a segment's rows touch
  const int* obs_seg_ob;
  int obs_nrows, obs_ncols; // block geometry (max over segments / strips)
  // FOTS by-products (nullable)
  FotsReduce* fots_part;    // [frame][fots_stride]: slot strip * nseg + seg, the rest filled with identity records
  int fots_stride;
  float* pix_z; uint8_t* pix_m; int n_markers;
  const int* mk_row_ptr;    // (H + 1,) CSR over frame rows
  const int* mk_x;          // marker column
  const int* mk_id;         // marker index
};

__device__ __forceinline__ float dpp_from_left(float v) {   // lane i receives lane i-1's value
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float v) {  // lane i receives lane i+1's value
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ int dpp_from_left_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_from_right_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }

// LDS traffic of ONE wave is executed in order; the fence only stops the compiler from moving accesses across it
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int... KS>
struct StreamCfg {
  static constexpr int NL = sizeof...(KS);
  static constexpr int K[NL] = {KS...};
  static constexpr int R(int l) { return (K[l] - 1) / 2; }
  static constexpr int sum_r() { int s = 0; for (int i = 0; i < NL; ++i) s += R(i); return s; }
  static constexpr int delay(int l) { int s = 0; for (int i = 0; i <= l; ++i) s += R(i); return s; }  // output row = y - delay
  static constexpr int acc_off(int l) { int s = 0; for (int i = 0; i < l; ++i) s += K[i] - 1; return s; }
  static constexpr int n_acc() { return acc_off(NL); }
  static constexpr int HALO = sum_r() + 1;                              // + the central-difference neighbour
  static constexpr int HL = (HALO + kStreamPx - 1) / kStreamPx;         // halo lanes per side
  static constexpr int VW = (64 - 2 * HL) * kStreamPx;                  // widest valid strip
  static_assert(sum_r() - R(NL - 1) < kStreamRing, "restore ring too shallow");
};

template <bool GZ, int... KS>
__global__ __launch_bounds__(64 * kStreamWaves) void taxim_stream_kernel(StreamArgs a) {
  using C = StreamCfg<KS...>;
  constexpr int NL = C::NL, PX = kStreamPx, SUMR = C::sum_r(), HL = C::HL;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wv_in_blk = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int per_frame = a.nstrips * a.nseg;
  const int wv = blockIdx.x * kStreamWaves + wv_in_blk;
  if (wv >= a.B * per_frame) return;
  const int frame = wv / per_frame;
  const int rem = wv - frame * per_frame;
  const int strip = rem / a.nseg, seg = rem - strip * a.nseg;
  const int H = a.H, W = a.W;
  const int r0 = seg * a.seg_rows, r1 = min(H, r0 + a.seg_rows);  // output rows of this wave
  const int vx0 = strip * a.strip_w, vx1 = min(W, vx0 + a.strip_w);
  const int cx0 = vx0 - HL * PX;                                   // column of lane 0, pixel 0
  // wave-private LDS: S ring [kStreamRing][64] float4 (.w unused) + one observation staging row [64 * PX * 3]
  v4f* ring = reinterpret_cast<v4f*>(smem_raw) + wv_in_blk * (kStreamRing * 64 + (64 * PX * 3) / 4);
  float* obs_row = reinterpret_cast<float*>(ring + kStreamRing * 64);

  const size_t fo = (size_t)frame * H * W;
  const float* __restrict__ zin = a.zin + fo;
  const float* __restrict__ hm = a.hm + fo;
  const float sa = a.shift_a[frame], sb = a.shift_b[frame];
  const float thr = -a.pdepth[frame] * a.contact_scale;  // TT:459

  int xg[PX];
  unsigned xo[PX];   // reflected + clamped column (loads)
  bool valid[PX];
  float X[PX];       // polynomial feature x of the pixel (TT:139-157)
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    xg[i] = cx0 + lane * PX + i;
    xo[i] = (unsigned)min(max(reflect_idx(xg[i], W), 0), W - 1);
    valid[i] = xg[i] >= vx0 && xg[i] < vx1;
    X[i] = a.sh.rgb ? a.sh.fx[min(max(xg[i], 0), W - 1)] : 0.0f;
  }

  // taps (wave-uniform: scalar registers)
  float w[C::n_acc() + NL];  // level l: w[acc_off(l) + l + t], t < K
  static_for<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    static_for<0, C::K[l]>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      w[C::acc_off(l) + l + t] = a.taps[l][t < C::K[l] - 1 - t ? t : C::K[l] - 1 - t];
    });
  });

  float A[C::n_acc() > 0 ? C::n_acc() : 1][PX];  // vertical partial sums
#pragma unroll
  for (int j = 0; j < C::n_acc(); ++j)
#pragma unroll
    for (int i = 0; i < PX; ++i) A[j][i] = 0.0f;
  float Zu[PX] = {0.f, 0.f, 0.f}, Zm[PX] = {0.f, 0.f, 0.f}, Zd[PX] = {0.f, 0.f, 0.f};  // last-level rows g-1, g, g+1

  // FOTS contact statistics of this wave's pixels
  float f_zmax = -INFINITY;
  int f_cnt = 0, f_sr = 0, f_sc = 0;
  // policy observation: vertical partial sums of the <= 3 observation rows in flight
  float OA[kStreamObsActive][PX * 3];
#pragma unroll
  for (int k = 0; k < kStreamObsActive; ++k)
#pragma unroll
    for (int j = 0; j < PX * 3; ++j) OA[k][j] = 0.0f;
  const bool do_obs = a.obs_part != nullptr && a.sh.rgb != nullptr;
  int cur_o0 = do_obs ? a.obs_seg_oa[seg] : 0;
  const int seg_ob = do_obs ? a.obs_seg_ob[seg] : -1;
  float* const obs_blk = do_obs ? a.obs_part + (size_t)wv * (a.obs_nrows * a.obs_ncols * 3) : nullptr;
  const int seg_oa = cur_o0;

  auto row_off = [&](int y) -> unsigned { return (unsigned)min(max(reflect_idx(y, H), 0), H - 1) * (unsigned)W; };

  // horizontal reduction of one finished (or segment-final) observation row: staged through the wave's LDS row
  auto obs_flush = [&](const float (&acc)[PX * 3], int o) {
    if (o < seg_oa || o > seg_ob) return;
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < PX * 3; ++j) obs_row[lane * (PX * 3) + j] = acc[j];
    wave_lds_fence();
    const int q0 = a.obs_strip_q0[strip], nq = a.obs_strip_nq[strip];
    for (int base = 0; base < nq * 3; base += 64) {
      const int j = base + lane;
      if (j < nq * 3) {
        const int qi = j / 3, ch = j - qi * 3, q = q0 + qi;
        const int xlo = a.obs_xlo[q], xhi = xlo + a.obs_xcnt[q];
        const int xa = max(xlo, vx0), xb = min(xhi, vx1);
        const float* wq = a.obs_wx + (size_t)q * a.obs_kx - xlo;
        float s = 0.0f;
        for (int x = xa; x < xb; ++x) s = fmaf(wq[x], obs_row[(x - cx0) * 3 + ch], s);
        obs_blk[((o - seg_oa) * a.obs_ncols + qi) * 3 + ch] = s;
      }
    }
    wave_lds_fence();
  };

  const int ys = r0 - SUMR - 1, ye = r1 - 1 + SUMR + 1;  // input rows walked by this wave
  float zc[PX], hc[PX];
  {
    const unsigned ro = row_off(ys);
#pragma unroll
    for (int i = 0; i < PX; ++i) { zc[i] = zin[ro + xo[i]]; hc[i] = hm[ro + xo[i]]; }
  }
  for (int y = ys; y <= ye; ++y) {
    // ---- prefetch the next input row ----
    float zn[PX], hn[PX];
    {
      const unsigned ro = row_off(y + 1 <= ye ? y + 1 : y);
#pragma unroll
      for (int i = 0; i < PX; ++i) { zn[i] = zin[ro + xo[i]]; hn[i] = hm[ro + xo[i]]; }
    }
    // ---- S of this row into the ring; contact statistics of the rows this wave owns ----
    float S[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) S[i] = (hc[i] - sa) - sb;  // TT:441
    ring[(y & (kStreamRing - 1)) * 64 + lane] = (v4f){S[0], S[1], S[2], 0.0f};
    if (a.fots_part != nullptr && y >= r0 && y < r1) {
      float gl[PX] = {0.f, 0.f, 0.f};
      if constexpr (!GZ) {
        const unsigned ro = row_off(y);
#pragma unroll
        for (int i = 0; i < PX; ++i) gl[i] = a.gel[ro + xo[i]];
      }
      int mrow[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        const float J = fmin_raw(S[i], gl[i]);
        const int m1 = (valid[i] && ((J - gl[i]) < thr) && (S[i] < 0.0f)) ? 1 : 0;  // TT:457-461
        mrow[i] = m1;
        f_cnt += m1; f_sr += m1 * y; f_sc += m1 * xg[i];
      }
      if (a.pix_m != nullptr) {  // contact mask at the FOTS marker pixels of this row
        for (int e = a.mk_row_ptr[y]; e < a.mk_row_ptr[y + 1]; ++e) {
          const int d = a.mk_x[e] - cx0 - lane * PX;
          if (d >= 0 && d < PX && a.mk_x[e] >= vx0 && a.mk_x[e] < vx1)
            a.pix_m[(size_t)frame * a.n_markers + a.mk_id[e]] = (uint8_t)(d == 0 ? mrow[0] : (d == 1 ? mrow[1] : mrow[2]));
        }
      }
    }
    // ---- the levels: horizontal pass over the lanes, vertical scatter into the partial sums, masked restore ----
    float cur[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) cur[i] = zc[i];
    static_for<0, NL>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      constexpr int K = C::K[l], R = C::R(l), WO = C::acc_off(l) + l, AO = C::acc_off(l);
      float h[PX];
      if constexpr (K == 1) {
#pragma unroll
        for (int i = 0; i < PX; ++i) h[i] = cur[i];
      } else {
        // window of pixel offsets -R .. PX - 1 + R around this lane's pixels, gathered from the neighbouring lanes
        constexpr int NS = (R + PX - 1) / PX;  // lanes needed on each side
        float win[PX + 2 * R];
        float L[PX], Rr[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) { L[i] = cur[i]; Rr[i] = cur[i]; win[R + i] = cur[i]; }
        static_for<1, NS + 1>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
#pragma unroll
          for (int i = 0; i < PX; ++i) { L[i] = dpp_from_left(L[i]); Rr[i] = dpp_from_right(Rr[i]); }
          static_for<0, PX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int pl = -k * PX + i;    // pixel offset of lane -k, pixel i
            constexpr int pr = k * PX + i;
            if constexpr (pl >= -R) win[R + pl] = L[i];
            if constexpr (pr <= PX - 1 + R) win[R + pr] = Rr[i];
          });
        });
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          float o = 0.0f;
          static_for<0, K>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            o = fmaf(w[WO + t], win[i + t], o);
          });
          h[i] = o;
        }
      }
      float out[PX];
      if constexpr (K == 1) {
#pragma unroll
        for (int i = 0; i < PX; ++i) out[i] = h[i];
      } else {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          out[i] = fmaf(w[WO + K - 1], h[i], A[AO + K - 2][i]);
          static_for<0, K - 2>([&](auto jc) {
            constexpr int j = K - 2 - decltype(jc)::value;  // K-2 .. 1
            A[AO + j][i] = fmaf(w[WO + j], h[i], A[AO + j - 1][i]);
          });
          A[AO][i] = w[WO] * h[i];
        }
      }
      if constexpr (l < NL - 1) {  // TT:467 Z[M] = J[M]; the final blur (TT:468-471) has no restore
        const int rr = y - C::delay(l);
        const v4f Sv = ring[(rr & (kStreamRing - 1)) * 64 + lane];
        float gl[PX] = {0.f, 0.f, 0.f};
        if constexpr (!GZ) {
          const unsigned ro = row_off(rr);
#pragma unroll
          for (int i = 0; i < PX; ++i) gl[i] = a.gel[ro + xo[i]];
        }
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          const float Si = Sv[i];
          const float J = fmin_raw(Si, gl[i]);
          out[i] = (((J - gl[i]) < thr) && (Si < 0.0f)) ? J : out[i];
        }
      }
#pragma unroll
      for (int i = 0; i < PX; ++i) cur[i] = out[i];
    });
    // ---- cur = last-level row y - SUMR ----
#pragma unroll
    for (int i = 0; i < PX; ++i) { Zu[i] = Zm[i]; Zm[i] = Zd[i]; Zd[i] = cur[i]; }
    const int zr = y - SUMR;  // row index of Zd
    if (zr >= r0 && zr < r1) {
      if (a.fots_part != nullptr) {
#pragma unroll
        for (int i = 0; i < PX; ++i) f_zmax = valid[i] ? fmaxf(f_zmax, cur[i]) : f_zmax;
      }
      if (a.pix_z != nullptr) {
        for (int e = a.mk_row_ptr[zr]; e < a.mk_row_ptr[zr + 1]; ++e) {
          const int d = a.mk_x[e] - cx0 - lane * PX;
          if (d >= 0 && d < PX && a.mk_x[e] >= vx0 && a.mk_x[e] < vx1)
            a.pix_z[(size_t)frame * a.n_markers + a.mk_id[e]] = d == 0 ? cur[0] : (d == 1 ? cur[1] : cur[2]);
        }
      }
    }
    // ---- shading of row g = y - SUMR - 1 from (Zu, Zm, Zd) = rows g-1, g, g+1.  Replicate padding of the gradient maps
    //      (TT:501-502): rows 0 / H-1 take the gradient of rows 1 / H-2 and are emitted together with them; columns 0 / W-1
    //      take the bins of columns 1 / W-2. ----
    const int g = zr - 1;
    if (a.sh.rgb != nullptr && g >= max(r0, 1) && g <= min(r1 - 1, H - 2)) {
      const float zl = dpp_from_left(Zm[PX - 1]), zrg = dpp_from_right(Zm[0]);
      int code[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        int im, id;
        shade_bins(a.sh, Zu[i], Zd[i], i == 0 ? zl : Zm[i - 1], i == PX - 1 ? zrg : Zm[i + 1], im, id);
        code[i] = im | (id << 8);
      }
      const int cl = dpp_from_left_i(code[PX - 1]), cr = dpp_from_right_i(code[0]);
      int codec[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        const int right = i == PX - 1 ? cr : code[i + 1], left = i == 0 ? cl : code[i - 1];
        codec[i] = xg[i] == 0 ? right : (xg[i] == W - 1 ? left : code[i]);
      }
      // rows emitted by this iteration (ascending)
      const int e_lo = g == 1 ? 0 : g, e_hi = g == H - 2 ? H - 1 : g;
      for (int e = e_lo; e <= e_hi; ++e) {
        if (e < r0 || e >= r1) continue;
        const float Y = a.sh.fy[e];
        float rgb[PX * 3];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          float c[3];
          shade_poly(a.sh, codec[i] & 0xff, codec[i] >> 8, X[i], Y, c);
          const unsigned p = (unsigned)e * (unsigned)W + (unsigned)min(max(xg[i], 0), W - 1);
          const float* __restrict__ bg = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.sh.bg) + p * 12u);
          rgb[3 * i + 0] = __builtin_amdgcn_fmed3f(c[0] + bg[0], 0.0f, 1.0f);  // TT:257-258
          rgb[3 * i + 1] = __builtin_amdgcn_fmed3f(c[1] + bg[1], 0.0f, 1.0f);
          rgb[3 * i + 2] = __builtin_amdgcn_fmed3f(c[2] + bg[2], 0.0f, 1.0f);
          if (valid[i]) {
            float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(a.sh.rgb + fo * 3) + p * 12u);
            o[0] = rgb[3 * i]; o[1] = rgb[3 * i + 1]; o[2] = rgb[3 * i + 2];
          }
        }
        if (do_obs) {
          const int o0 = a.obs_row_o0[e];
          while (cur_o0 < o0) {  // the oldest observation row in flight got its last frame row: reduce it horizontally
            obs_flush(OA[0], cur_o0);
#pragma unroll
            for (int j = 0; j < PX * 3; ++j) { OA[0][j] = OA[1][j]; OA[1][j] = OA[2][j]; OA[2][j] = 0.0f; }
            ++cur_o0;
          }
          const float w0 = a.obs_row_w[e * 3], w1 = a.obs_row_w[e * 3 + 1], w2 = a.obs_row_w[e * 3 + 2];
#pragma unroll
          for (int j = 0; j < PX * 3; ++j) {
            OA[0][j] = fmaf(w0, rgb[j], OA[0][j]);
            OA[1][j] = fmaf(w1, rgb[j], OA[1][j]);
            OA[2][j] = fmaf(w2, rgb[j], OA[2][j]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) { zc[i] = zn[i]; hc[i] = hn[i]; }
  }
  if (do_obs) {  // observation rows still in flight at the end of the segment (another segment adds its share)
    obs_flush(OA[0], cur_o0);
    obs_flush(OA[1], cur_o0 + 1);
    obs_flush(OA[2], cur_o0 + 2);
    // rows of the block this segment never reached stay unwritten: the finishing kernel only reads [seg_oa, seg_ob]
  }
  if (a.fots_part != nullptr) {  // one record per wave: no atomics; fots_combine_kernel adds the records of an env
    f_zmax = wave_scan_max_lane63(f_zmax);
    f_cnt = wave_scan_add_lane63(f_cnt);
    f_sr = wave_scan_add_lane63(f_sr);
    f_sc = wave_scan_add_lane63(f_sc);
    if (lane == 63) {
      FotsReduce r;
      r.zmax = f_zmax; r.count = f_cnt; r.sum_row = f_sr; r.sum_col = f_sc;
      a.fots_part[(size_t)frame * a.fots_stride + rem] = r;
    }
    if (rem == 0) {  // the consumer adds fots_stride records per env: identity records for the slots no wave owns
      FotsReduce id;
      id.zmax = -INFINITY; id.count = 0; id.sum_row = 0; id.sum_col = 0;
      for (int s2 = per_frame + lane; s2 < a.fots_stride; s2 += 64) a.fots_part[(size_t)frame * a.fots_stride + s2] = id;
    }
  }
}

// adds the per-(strip, segment) partial sums of every observation cell in a fixed order and normalises by the weight sums
template <bool U8>
__global__ __launch_bounds__(256) void obs_finish_stream_kernel(const float* __restrict__ part, void* __restrict__ obs_v, ObsTables T,
                                                               const int* __restrict__ strip_q0, const int* __restrict__ strip_nq,
                                                               const int* __restrict__ seg_oa, const int* __restrict__ seg_ob,
                                                               int nstrips, int nseg, int nrows, int ncols) {
  const int oh = T.oh, ow = T.ow;
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= oh * ow) return;
  const int b = blockIdx.y;
  const int oy = cell / ow, ox = cell - oy * ow;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  for (int s = 0; s < nstrips; ++s) {
    const int qi = ox - strip_q0[s];
    if (qi < 0 || qi >= strip_nq[s]) continue;
    for (int g = 0; g < nseg; ++g) {
      if (oy < seg_oa[g] || oy > seg_ob[g]) continue;
      const size_t blk = ((size_t)b * nstrips + s) * nseg + g;
      const float* pp = part + blk * ((size_t)nrows * ncols * 3) + ((size_t)(oy - seg_oa[g]) * ncols + qi) * 3;
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

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int... KS>
static bool stream_geometry_t(int W, int* nstrips, int* strip_w) {
  using C = StreamCfg<KS...>;
  const int ns = (W + C::VW - 1) / C::VW;
  *nstrips = ns;
  *strip_w = (W + ns - 1) / ns;
  return true;
}

// which fused level sets have a streaming instantiation (the same sets the tiled tail covers)
static int stream_variant(int n_fused, int k0) {
  if (n_fused == 4 && k0 == 9) return 0;   // <9,5,3,5>  320x240
  if (n_fused == 3 && k0 == 9) return 1;   // <9,5,9>    640x480 (k = 15 stays a band level)
  return -1;
}

bool stream_supported(int n_fused, int k0, int H, int W) {
  static const int en = getenv("TACEX_TAIL_STREAM") ? atoi(getenv("TACEX_TAIL_STREAM")) : 1;
  return en != 0 && stream_variant(n_fused, k0) >= 0 && H >= 16 && W >= 16;
}

bool stream_geometry(int n_fused, int k0, int W, int* nstrips, int* strip_w) {
  switch (stream_variant(n_fused, k0)) {
    case 0: return stream_geometry_t<9, 5, 3, 5>(W, nstrips, strip_w);
    case 1: return stream_geometry_t<9, 5, 9>(W, nstrips, strip_w);
  }
  return false;
}

// vertical segments per strip: enough waves to give every SIMD ~2 of them, at most kStreamMaxSeg, each >= 24 rows
int stream_segments(int B, int nstrips, int H, int sum_r) {
  static const int forced = getenv("TACEX_STREAM_SEGS") ? atoi(getenv("TACEX_STREAM_SEGS")) : 0;
  int nseg = forced > 0 ? forced : (2048 + B * nstrips - 1) / (B * nstrips);
  if (nseg < 1) nseg = 1;
  if (nseg > kStreamMaxSeg) nseg = kStreamMaxSeg;
  while (nseg > 1 && (H / nseg < 24 || H - (nseg - 1) * ((H + nseg - 1) / nseg) < 4)) --nseg;
  (void)sum_r;
  return nseg;
}

template <int... KS>
static hipError_t launch_stream(const StreamArgs& a, bool gel_zero, hipStream_t st) {
  const int waves = a.B * a.nstrips * a.nseg;
  const dim3 grid((waves + kStreamWaves - 1) / kStreamWaves);
  const size_t lds = (size_t)kStreamWaves * (kStreamRing * 64 * 16 + 64 * kStreamPx * 3 * 4);
  if (gel_zero) hipLaunchKernelGGL((taxim_stream_kernel<true, KS...>), grid, dim3(64 * kStreamWaves), lds, st, a);
  else hipLaunchKernelGGL((taxim_stream_kernel<false, KS...>), grid, dim3(64 * kStreamWaves), lds, st, a);
  return hipGetLastError();
}

hipError_t run_stream_tail(const LevelDesc* lv, int n_levels, int n_fused, const float* zin, const float* hm, const float* gel,
                           const float* sa, const float* sb, const float* pd, const ShadeParams* sp, float* rgb, int B, int H, int W,
                           float contact_scale, const StreamPlan& plan, float* obs_part, FotsReduce* fots_part, int fots_stride,
                           float* pix_z, uint8_t* pix_m, hipStream_t st) {
  StreamArgs a{};
  a.zin = zin; a.hm = hm; a.gel = lv[0].gel_zero ? nullptr : gel; a.shift_a = sa; a.shift_b = sb; a.pdepth = pd;
  a.H = H; a.W = W; a.B = B; a.contact_scale = contact_scale;
  for (int i = 0; i < n_fused; ++i) a.taps[i] = lv[n_levels - n_fused + i].taps_w_dev;
  a.sh.poly = sp->poly_dev; a.sh.bg = sp->bg_nhwc_dev; a.sh.fx = sp->fx_dev; a.sh.fy = sp->fy_dev; a.sh.rgb = rgb;
  a.sh.idx_out = nullptr; a.sh.H = H; a.sh.W = W; a.sh.B = B; a.sh.nb = sp->nb; a.sh.pixmm = sp->pixmm;
  a.sh.calib_h = (float)sp->calib_h; a.sh.calib_w = (float)sp->calib_w; a.sh.x_binr = sp->x_binr; a.sh.y_binr = sp->y_binr;
  a.sh.gsy = (float)(0.5 * H / sp->calib_h / (double)sp->pixmm); a.sh.gsx = (float)(0.5 * W / sp->calib_w / (double)sp->pixmm);
  a.sh.inv_x_binr = (float)(1.0 / (double)sp->x_binr); a.sh.inv_y_binr = (float)(1.0 / (double)sp->y_binr);
  a.nstrips = plan.nstrips; a.strip_w = plan.strip_w; a.nseg = plan.nseg; a.seg_rows = plan.seg_rows;
  if (obs_part && plan.obs_ready) {
    a.obs_part = obs_part;
    a.obs_row_o0 = plan.obs_row_o0; a.obs_row_w = plan.obs_row_w;
    a.obs_xlo = plan.obs.xlo; a.obs_xcnt = plan.obs.xcnt; a.obs_wx = plan.obs.wx; a.obs_kx = plan.obs.kx;
    a.obs_strip_q0 = plan.obs_strip_q0; a.obs_strip_nq = plan.obs_strip_nq;
    a.obs_seg_oa = plan.obs_seg_oa; a.obs_seg_ob = plan.obs_seg_ob;
    a.obs_nrows = plan.obs_nrows; a.obs_ncols = plan.obs_ncols;
  }
  a.fots_part = fots_part; a.fots_stride = fots_stride;
  if (pix_z && pix_m && plan.mk_row_ptr) {
    a.pix_z = pix_z; a.pix_m = pix_m; a.n_markers = plan.n_markers;
    a.mk_row_ptr = plan.mk_row_ptr; a.mk_x = plan.mk_x; a.mk_id = plan.mk_id;
  }
  const int k0 = lv[n_levels - n_fused].kw;
  switch (stream_variant(n_fused, k0)) {
    case 0: return launch_stream<9, 5, 3, 5>(a, lv[0].gel_zero, st);
    case 1: return launch_stream<9, 5, 9>(a, lv[0].gel_zero, st);
  }
  return hipErrorInvalidValue;
}

hipError_t run_obs_finish_stream(const float* part, void* obs, bool u8, const StreamPlan& plan, int B, hipStream_t st) {
  const dim3 grid((plan.obs.oh * plan.obs.ow + 255) / 256, B);
  if (u8)
    hipLaunchKernelGGL(obs_finish_stream_kernel<true>, grid, dim3(256), 0, st, part, obs, plan.obs, plan.obs_strip_q0, plan.obs_strip_nq,
                       plan.obs_seg_oa, plan.obs_seg_ob, plan.nstrips, plan.nseg, plan.obs_nrows, plan.obs_ncols);
  else
    hipLaunchKernelGGL(obs_finish_stream_kernel<false>, grid, dim3(256), 0, st, part, obs, plan.obs, plan.obs_strip_q0, plan.obs_strip_nq,
                       plan.obs_seg_oa, plan.obs_seg_ob, plan.nstrips, plan.nseg, plan.obs_nrows, plan.obs_ncols);
  return hipGetLastError();
}

}  // namespace tacex
