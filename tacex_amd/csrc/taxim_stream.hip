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
//   * input rows never pass through registers: one global_load_lds per array lands a row in a two-slot LDS ring a full iteration
//     before it is read back; the iteration has ONE vmcnt wait (mid_point), behind which only the RGB stores are issued - nothing
//     ever waits for a young load or store (gfx9 stores count in vmcnt, and the counter retires in order);
//   * what the frame's contact row range (frame_rows_kernel) rules out is not done: the height map is loaded on contact rows
//     only, level rows only within the band levels' reach, and a wave no contact can reach shades a flat gel (no levels, no bins).
// The kernel is bound by the ISSUE of its vector-memory instructions (9 + <= 15 table-gather instructions per row, ~50 % of
// the time against a 481 us compute floor at 1024 frames) and by f32 VALU issue - see DESIGN.md section 4.2 for the probes.
// HBM sees one read of the previous level (+ the height map on contact rows) and one write of RGB; the halo is only
// horizontal (2 x 12 of 184 columns, those re-reads are L2 hits) plus sum(R) + 1 warm-up rows per vertical segment.
//
// Summation order of every level equals the band kernels / the tiled tail (taps ascending, fmaf chains), so the deformed gel
// is bit-identical to theirs.  Reference semantics: TT:464-471 (levels + restore), TT:411 (reflect padding: the strip loads
// rows / columns at REFLECTED coordinates, a symmetric kernel keeps the halo mirror-symmetric), TT:475-503, 237-258 (shading).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "tacex_internal.h"
#include "taxim_device.h"

namespace tacex {

constexpr int kStreamPx = 3;         // pixels per lane
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
  float mag0_t2;         // squared gradient magnitude below which the magnitude bin is 0 for certain (see shade_part1)
  int nstrips, strip_w;  // strips per frame, valid columns per strip
  int nseg, seg_rows;    // vertical segments per strip, rows per segment
  const int* rows;       // (H, kStreamRowInts): StreamRowInfo record + packed marker slots per frame row
  int mk_vec;            // marker slots usable (every row has <= kStreamMkSlots markers)
  // policy observation (nullable): per (frame, strip, segment) block of partial sums [obs_nrows][obs_ncols][3]
  float* obs_part;
  const int* obs_xlo; const int* obs_xcnt; const float* obs_wx; int obs_kx;   // column filters (ObsTables)
  const int* obs_strip_q0;  // (nstrips,) first observation column a strip's valid columns touch
  const int* obs_strip_nq;  // (nstrips,) number of such columns
  const int* obs_seg_oa;    // (nseg,) first / last observation row a segment's rows touch
  const int* obs_seg_ob;
  int obs_nrows, obs_ncols; // block geometry (max over segments / strips)
  int obs_kxp;              // column-filter taps padded to a multiple of 4 (<= kStreamObsLdsFloats / obs_ncols)
  // FOTS by-products (nullable)
  FotsReduce* fots_part;    // [frame][fots_stride]: slot strip * nseg + seg, the rest filled with identity records
  int fots_stride;
  float* z_out;             // levels role: (B,H,W) last level
  float* pix_z; uint8_t* pix_m; int n_markers;
  const int* mk_x;          // marker column  (CSR over rows: StreamRowInfo::mk0 / mk1)
  const int* mk_id;         // marker index
  const int* rows_ext;      // (B,4) contact row range | contact column range of every frame (frame_rows_kernel), nullable
  int ext_grow;             // rows by which the band levels have spread the non-zero range of zin beyond it
  const int* order;         // (B * nstrips * nseg) item of every launched wave, heaviest first (stream_order_kernel); nullptr: identity
  const float* flat_rgb;    // (H,W,3) RGB of the undeformed gel (stream_flat_image_kernel); nullptr: flat rows evaluate the polynomial
};

__device__ __forceinline__ float dpp_from_left(float v) {   // lane i receives lane i-1's value
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));  // no `old` operand to initialise
}
__device__ __forceinline__ float dpp_from_right(float v) {  // lane i receives lane i+1's value
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ int dpp_from_left_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_from_right_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }

// d = w * h + a with the destination free to differ from the addend: the vertical scatter chain writes A[j] from A[j-1], and
// hipcc's two-address v_fmac form (destination tied to the addend) costs one v_mov per partial sum and row to put the result
// back into the loop-carried register.  volatile: the chain must run in descending j (A[j] is read before it is overwritten).
__device__ __forceinline__ float fma_to(float w, float h, float a) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(w), "v"(h), "v"(a));
  return d;
}
// A wave-uniform value moved into a VECTOR register on purpose: on gfx950 a v_fma_f32 with a scalar-register operand issues at
// ~4.8 cycles per SIMD against ~3.0 with three vector operands (scripts/hip_probes/valu_occupancy.hip), and nearly every FMA of
// the levels multiplies by a tap.  The asm hides the uniformity from hipcc, which would otherwise fold the scalar back in.
__device__ __forceinline__ float to_vgpr(float s) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}

// LDS traffic of ONE wave is executed in order; the fence only stops the compiler from moving accesses across it
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef float v3f __attribute__((ext_vector_type(3)));

constexpr int kStreamObsLdsFloats = 448;
constexpr int kStreamObsMaxCols = 64;
constexpr size_t kStreamStageBytes = 64 * kStreamPx * 3 * 4;  // one row of 3 floats per pixel (background in / RGB out)
// workgroup-shared copy of the polynomial records of magnitude bin 0 ([id][20] floats = the five 16-byte pieces the polynomial
// reads: an ODD number of 16-byte bank slots per record, so lanes with consecutive direction bins read distinct slots).  Nearly all pixels outside a contact - and there the direction
// bin still varies from pixel to pixel - gather their record from here instead of through the vector L1 (the table gathers were
// 2/3 of the fused kernel's TCP_TOTAL_CACHE_ACCESSES: 636 per row and wave).
constexpr int kStreamPolyPitch = 20;
constexpr int kStreamPolyMaxBins = 128;
constexpr size_t kStreamLdsShared = (size_t)kStreamPolyMaxBins * kStreamPolyPitch * 4;

enum { kStreamFused = 0, kStreamLevels = 1, kStreamShade = 2 };
#ifndef TACEX_STREAM3_WAVES
#define TACEX_STREAM3_WAVES 3
#endif

// ROLE: kStreamFused  - levels + shading in one kernel (one wave keeps ~240 registers: two waves per SIMD) - the default
//       kStreamLevels - levels only: writes the last level (B,H,W) + the FOTS by-products; lean (four waves per SIMD)
//       kStreamShade  - shading only (KS empty): reads the last level, writes RGB + observation partial sums
// The split pair (TACEX_STREAM_SPLIT=1) is an A/B path: it costs 8 B/px of extra traffic (the last level is written and re-read
// once) and measured slower than the fused kernel (70 + 230 us vs 215 us per 256 frames).
template <int ROLE, int... KS>
struct StreamCfg {
  static constexpr int NL = sizeof...(KS);
  static constexpr int K[NL > 0 ? NL : 1] = {KS...};
  static constexpr bool SHADE = ROLE != kStreamLevels;
  static constexpr int R(int l) { return (K[l] - 1) / 2; }
  static constexpr int sum_r() { int s = 0; for (int i = 0; i < NL; ++i) s += R(i); return s; }
  static constexpr int delay(int l) { int s = 0; for (int i = 0; i <= l; ++i) s += R(i); return s; }  // output row = y - delay
  static constexpr int acc_off(int l) { int s = 0; for (int i = 0; i < l; ++i) s += K[i] - 1; return s; }
  static constexpr int n_acc() { return acc_off(NL); }
  static constexpr int HALO = sum_r() + (SHADE ? 1 : 0);                // + the central-difference neighbour
  static constexpr int HL = (HALO + kStreamPx - 1) / kStreamPx;         // halo lanes per side
  static constexpr int VW = (64 - 2 * HL) * kStreamPx;                  // widest valid strip
  static constexpr int last_r() { return NL > 0 ? R(NL - 1) : 0; }
  // waves per SIMD the kernel is compiled for: the lean levels kernel 4 (<= 128 VGPRs), the shading kernel 3 (<= 168), the
  // fused kernel 2 with four levels and TACEX_STREAM3_WAVES (3) with the three-level tail <5,3,5> (k = 9 left to a band level:
  // 24 fewer partial sums, a 12-row instead of a 20-row warm-up)
  static constexpr int min_waves() { return ROLE == kStreamLevels ? 4 : (ROLE == kStreamShade ? 3 : (sum_r() <= 5 ? TACEX_STREAM3_WAVES : 2)); }
  // S ring of the masked restores: the deepest restore (level NL-2) looks back sum(R) - R_last rows, and an iteration READS its
  // rows before it WRITES its own, so exactly that many rows are kept.  Two-wave kernels keep 16-byte records (one ds_read_b128
  // per restore); the three-wave kernel packs 12-byte records (read2_b32 + b32) to fit three workgroups into a CU's 160 KB.
  static constexpr int ring_rows() { return sum_r() - last_r() > 0 ? sum_r() - last_r() : 1; }
  static constexpr bool ring_packed() { return min_waves() >= 3; }
  // wave-private LDS: S ring [ring_rows][64] (restores) | observation staging row | column-filter window weights
  // [obs_ncols][obs_kxp] | window start per column
  static constexpr size_t ring_bytes() { return NL > 1 ? (size_t)ring_rows() * 64 * (ring_packed() ? 12 : 16) : 0; }
  static constexpr size_t shade_bytes() { return SHADE ? kStreamStageBytes + kStreamObsLdsFloats * 4 + kStreamObsMaxCols * 4 : 0; }
  // input rows land in a two-slot ring straight from memory (global_load_lds_dwordx3: 16-byte lane stride): [slot][z | hm][64 x 4]
  static constexpr size_t rows_bytes() { return (size_t)2 * (NL > 0 ? 2 : 1) * 64 * 16; }
  static constexpr size_t lds_per_wave() { return ring_bytes() + shade_bytes() + rows_bytes(); }
  static constexpr size_t lds_shared() { return SHADE ? kStreamLdsShared : 0; }
  static constexpr size_t lds_bytes() { return lds_shared() + kStreamWaves * lds_per_wave(); }
  static_assert(lds_per_wave() % 16 == 0, "wave-private LDS blocks must stay 16-byte aligned");
  static_assert(min_waves() < 3 || 3 * lds_bytes() <= 160 * 1024, "three workgroups per CU need <= 53.3 KB of LDS each");
};

struct __attribute__((packed, aligned(4))) StreamS3 { float x, y, z; };

// the 18 coefficients of one table record: four 16-byte pieces and one 8-byte piece (18 registers, not 20)
struct StreamRec { v4f c0, c1, c2, c3; v2f c4; };
__device__ __forceinline__ StreamRec stream_rec_load(const float* p) {  // p 16-byte aligned (global table or the LDS copy)
  StreamRec r;
  const v4f* q = reinterpret_cast<const v4f*>(p);
  r.c0 = q[0]; r.c1 = q[1]; r.c2 = q[2]; r.c3 = q[3];
  r.c4 = *reinterpret_cast<const v2f*>(p + 16);
  return r;
}
// I_c = sum_k f_k(X, Y) p_{c,k}, f = [X^2, Y^2, XY, X, Y, 1] (TT:148-157, 250-255)
__device__ __forceinline__ void stream_poly(float X, float Y, const StreamRec& r, float& p0, float& p1, float& p2) {
  const v4f c0 = r.c0, c1 = r.c1, c2 = r.c2, c3 = r.c3;
  const v2f c4 = r.c4;
  const float f0 = X * X, f1 = Y * Y, f2 = X * Y;
  p0 = fmaf(f0, c0.x, fmaf(f1, c0.y, fmaf(f2, c0.z, fmaf(X, c0.w, fmaf(Y, c1.x, c1.y)))));
  p1 = fmaf(f0, c1.z, fmaf(f1, c1.w, fmaf(f2, c2.x, fmaf(X, c2.y, fmaf(Y, c2.z, c2.w)))));
  p2 = fmaf(f0, c3.x, fmaf(f1, c3.y, fmaf(f2, c3.z, fmaf(X, c3.w, fmaf(Y, c4.x, c4.y)))));
}

// MINW = StreamCfg<ROLE, KS...>::min_waves() (a pack cannot be expanded inside the launch-bounds attribute)
template <bool GZ, int ROLE, int MINW, int... KS>
__global__ __launch_bounds__(64 * kStreamWaves, MINW) void taxim_stream_kernel(StreamArgs a) {
  using C = StreamCfg<ROLE, KS...>;
  constexpr int NL = C::NL, PX = kStreamPx, SUMR = C::sum_r(), HL = C::HL;
  constexpr bool SHADE = C::SHADE, LEVELS = NL > 0;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wv_in_blk = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int per_frame = a.nstrips * a.nseg;
  float* const polyL = reinterpret_cast<float*>(smem_raw);
  int nb_lds = 0;  // table records [0, nb_lds) are LDS-resident
  if constexpr (SHADE) {  // magnitude-bin-0 records -> LDS (all waves of the workgroup, before any of them may leave)
    const int nb = a.sh.nb;
    if (nb <= kStreamPolyMaxBins) {
      nb_lds = nb;
      for (int j = threadIdx.x; j < nb * 5; j += 64 * kStreamWaves) {  // 5 x 16 bytes per record (floats 18, 19 are padding)
        const int id = j / 5, k = j - id * 5;
        *reinterpret_cast<v4f*>(polyL + id * kStreamPolyPitch + 4 * k) = *reinterpret_cast<const v4f*>(a.sh.poly + (size_t)id * 24 + 4 * k);
      }
    }
    __syncthreads();
  }
  // Work distribution.  An item = one (frame, strip, row segment) = one wave.  Items differ by up to 3x (flat waves; rows
  // through a contact gather table records and run the magnitude arc tangent), a workgroup slot stays occupied until the
  // slowest of its four waves is done, and with items in frame order 27 % of the wave-slot time of a launch was idle
  // (SQ_WAVE_CYCLES against GRBM_GUI_ACTIVE, profiles/r03_experiments.md).  stream_order_kernel therefore sorts the items by
  // the number of contact rows in their segment, heaviest first: the four waves of a workgroup get items of equal weight and
  // the light ones fill the tail of the launch (longest-processing-time-first).
  const int n_items = a.B * per_frame;
  const int slot = blockIdx.x * kStreamWaves + wv_in_blk;
  if (slot >= n_items) return;
  const int wv = a.order != nullptr ? __builtin_amdgcn_readfirstlane(a.order[slot]) : slot;
  const int frame = wv / per_frame;
  const int rem = wv - frame * per_frame;
  const int strip = rem / a.nseg, seg = rem - strip * a.nseg;
  const int H = a.H, W = a.W;
  const int r0 = seg * a.seg_rows, r1 = min(H, r0 + a.seg_rows);  // output rows of this wave
  const int vx0 = strip * a.strip_w, vx1 = min(W, vx0 + a.strip_w);
  const int cx0 = vx0 - HL * PX;                                   // column of lane 0, pixel 0
  char* lds = smem_raw + C::lds_shared() + (size_t)wv_in_blk * C::lds_per_wave();
  v4f* ring = reinterpret_cast<v4f*>(lds);
  StreamS3* ring3 = reinterpret_cast<StreamS3*>(lds);
  constexpr int NRING = C::ring_rows();
  float* obs_row = reinterpret_cast<float*>(lds + C::ring_bytes());  // [64 * PX * 3] observation staging row
  float* obs_wl = obs_row + 64 * PX * 3;                              // [nq][kxp] window weights
  int* obs_xb = reinterpret_cast<int*>(obs_wl + kStreamObsLdsFloats);  // [nq] first staging pixel of the window
  float* const rowbuf = reinterpret_cast<float*>(lds + C::ring_bytes() + C::shade_bytes());  // [2 slots][z | hm][64 lanes x 4]
  constexpr int kRowArr = 64 * 4, kRowSlot = (NL > 0 ? 2 : 1) * kRowArr;                     // floats

  const size_t fo = (size_t)frame * H * W;
  const float* __restrict__ zin = a.zin + fo;
  const float* __restrict__ hm = a.hm + fo;
  float sa = 0.0f, sb = 0.0f, thr = 0.0f;
  if constexpr (LEVELS) {
    sa = a.shift_a[frame]; sb = a.shift_b[frame];
    thr = -a.pdepth[frame] * a.contact_scale;  // TT:459
  }
  // Zero gel map (GZ): J = min(S, 0) and the contact test (J - 0 < thr) && (S < 0) (TT:457-461) collapse to S < min(thr, 0)
  // with J = S on the mask - one compare per pixel instead of min + two compares + and (v_min / v_cmp issue at 2/3 rate).
  const float thr_eff = fminf(thr, 0.0f);
  const float mag0_t2 = a.mag0_t2;
  const bool do_obs = SHADE && a.obs_part != nullptr;
  const bool do_fots = LEVELS && a.fots_part != nullptr;

  int xg[PX];
  unsigned xo[PX];   // reflected + clamped column (loads)
  bool valid[PX];
  float X[PX];       // polynomial feature x of the pixel (TT:139-157)
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    xg[i] = cx0 + lane * PX + i;
    xo[i] = (unsigned)min(max(reflect_idx(xg[i], W), 0), W - 1);
    valid[i] = xg[i] >= vx0 && xg[i] < vx1;
    X[i] = SHADE ? a.sh.fx[min(max(xg[i], 0), W - 1)] : 0.0f;
  }
  // Input rows never pass through registers on their way in: each lane issues ONE 12-byte global_load_lds per array (the lanes
  // hanging over the image edge at a clamped address), the row lands in the wave's two-slot LDS ring and is read back one
  // iteration later - every pixel from the ring position that holds its (reflected) column, so the reflect padding costs no
  // extra loads and no merge.  The landing zone being LDS (not 12 VGPRs per row in flight) is what lets a row be issued a full
  // iteration ahead of the ONE vmcnt wait of the iteration (mid_point below): plain loads consumed at the loop tail were forced
  // early by that wait (in-order vmcnt) with ~0.6 iterations to cover the HBM latency.
  const unsigned xb = (unsigned)min(max(xg[0], 0), W - PX);
  int ridx[PX];  // ring position (float index inside one array of a slot) of this lane's pixels
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    const int c = (int)xo[i];
    const int lc = min(max((c - cx0) / PX, 0), 63);
    const int xbc = min(max(cx0 + lc * PX, 0), W - PX);
    ridx[i] = lc * 4 + min(max(c - xbc, 0), PX - 1);
  }
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // The height map only serves S = (hm - min) - press of the masked restore and the contact statistics, and S < 0 occurs on the
  // frame's contact rows alone (frame_rows_kernel): every other row skips its height-map load (one of the two row loads) and
  // reads S = +inf - no restore, no contact.
  // (-4.8 % on the kernel in alternating runs on one box.)  Likewise the previous level is zero beyond ext_grow rows of them.
  int hm_lo = 0, hm_hi = H - 1, z_lo = 0, z_hi = H - 1;
  if constexpr (GZ && LEVELS) {
#ifndef TACEX_STREAM_HM_ALWAYS  // (A/B probe: load both arrays of every row)
    if (a.rows_ext != nullptr) {
      hm_lo = a.rows_ext[4 * frame]; hm_hi = a.rows_ext[4 * frame + 1];
      z_lo = hm_lo - a.ext_grow; z_hi = hm_hi + a.ext_grow;
    }
#endif
  }
  auto issue_row = [&](int row, int slot) {
#ifdef TACEX_DBG_NO_ROWLOAD
    const unsigned ro = xb;
#else
    const unsigned ro = (unsigned)row * (unsigned)W + xb;
#endif
    float* dst = rowbuf + slot * kRowSlot;
    if (row >= z_lo && row <= z_hi) __builtin_amdgcn_global_load_lds((gptr_t)(zin + ro), (lptr_t)dst, 12, 0, 0);
    if constexpr (LEVELS)
      if (row >= hm_lo && row <= hm_hi) __builtin_amdgcn_global_load_lds((gptr_t)(hm + ro), (lptr_t)(dst + kRowArr), 12, 0, 0);
  };
  // interior lanes read back their own 16 bytes (conflict-free b128); only the few lanes with a pixel outside the image pick
  // their mirrored columns one by one (scattered b32 reads at a 4-dword lane stride would be 4-way bank conflicts for everyone)
  const bool border = xg[0] < 0 || xg[PX - 1] >= W;
  const int edge_lane = (W - 1 - cx0) / PX, edge_slot = (W - 1 - cx0) - edge_lane * PX;  // lane / pixel slot of image column W - 1
  auto read_row = [&](int slot, int row, float (&zz)[PX], float (&hh)[PX]) {
    const float* src = rowbuf + slot * kRowSlot;
    const bool have_hm = LEVELS && row >= hm_lo && row <= hm_hi, have_z = row >= z_lo && row <= z_hi;
    // All LDS reads of the row are issued back to back and waited for ONCE (a slot that was not loaded for this row holds an
    // older row: finite data, discarded by the wave-uniform selects below).  As first written - reads inside `if (have_z)` /
    // `if (have_hm)` / per-pixel border branches - hipcc waited after every single read: 630 of the 4 600 cycles of an iteration
    // (in-kernel clock, profiles/r03_experiments.md section 5).
    const v4f zq = *reinterpret_cast<const v4f*>(src + lane * 4);
    v4f hq = (v4f)(0.0f);
    if constexpr (LEVELS) hq = *reinterpret_cast<const v4f*>(src + kRowArr + lane * 4);
    float bz[PX] = {zq.x, zq.y, zq.z}, bh[PX] = {hq.x, hq.y, hq.z};
    if (border) {
#pragma unroll
      for (int i = 0; i < PX; ++i) bz[i] = src[ridx[i]];
      if constexpr (LEVELS) {
#pragma unroll
        for (int i = 0; i < PX; ++i) bh[i] = src[kRowArr + ridx[i]];
      }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      zz[i] = have_z ? bz[i] : 0.0f;
      if constexpr (LEVELS) hh[i] = have_hm ? bh[i] : INFINITY;
    }
  };

  // background of one row: 12 bytes per pixel straight into the lane that owns the pixel.  (Staging the row through LDS as
  // contiguous 16-byte pieces cut the L1 tag look-ups by 40 % but added four LDS round trips to the dependency chain of every
  // row: measured slower; perfectly linear 768-byte loads + stores gain at most 10 %)
  unsigned xc[PX];  // BYTE offset of the pixel's (clamped) column inside an RGB / background row: row * W * 12 is wave-uniform (scalar
                    // unit), so an address costs one v_add instead of v_add + v_mul_lo (2/3 rate) per pixel slot, load and store
                    // (nine v_mul_lo fewer per row; the kernel time did not move: 774-788 vs 783-789 us alternating on one box)
#pragma unroll
  for (int i = 0; i < PX; ++i) xc[i] = (unsigned)min(max(xg[i], 0), W - 1) * 12u;
  auto load_bg = [&](int row, v3f (&q)[PX]) {
#ifdef TACEX_DBG_NO_BG
    if (row >= 0) return;
#endif
#pragma unroll
    for (int i = 0; i < PX; ++i)
      q[i] = *reinterpret_cast<const v3f*>(reinterpret_cast<const char*>(a.sh.bg) + ((unsigned)row * (unsigned)W * 12u + xc[i]));
  };
  v3f bgq[PX] = {(v3f)(0.0f), (v3f)(0.0f), (v3f)(0.0f)};   // background of the row shaded in the current iteration

  // ---- policy observation set-up: this strip's column filters as fixed-length windows over the staging row ----
  int nq = 0;
  const int kxp = a.obs_kxp;
  if constexpr (SHADE) {
    if (do_obs) {
      const int q0 = a.obs_strip_q0[strip];
      nq = a.obs_strip_nq[strip];
      for (int j = lane; j < nq; j += 64) {
        const int q = q0 + j;
        int xb2 = max(a.obs_xlo[q], vx0) - cx0;
        xb2 = min(xb2, 64 * PX - kxp);
        obs_xb[j] = xb2;
      }
      wave_lds_fence();
      for (int j = lane; j < nq * kxp; j += 64) {
        const int qi = j / kxp, t = j - qi * kxp, q = q0 + qi;
        const int x = obs_xb[qi] + t + cx0;  // frame column of this tap
        const int xlo = a.obs_xlo[q];
        obs_wl[j] = (x >= xlo && x < xlo + a.obs_xcnt[q] && x >= vx0 && x < vx1) ? a.obs_wx[(size_t)q * a.obs_kx + (x - xlo)] : 0.0f;
      }
      wave_lds_fence();
    }
  }

  // taps (wave-uniform: scalar registers)
  float w[C::n_acc() + NL + 1];  // level l: w[acc_off(l) + l + t], t < K
  static_for<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    static_for<0, C::K[l]>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int ts = t < C::K[l] - 1 - t ? t : C::K[l] - 1 - t;  // symmetric taps share a register
      if constexpr (ts == t) w[C::acc_off(l) + l + t] = a.taps[l][t];
      else w[C::acc_off(l) + l + t] = w[C::acc_off(l) + l + ts];
    });
  });

  float A[C::n_acc() > 0 ? C::n_acc() : 1][PX];  // vertical partial sums
#pragma unroll
  for (int j = 0; j < C::n_acc(); ++j)
#pragma unroll
    for (int i = 0; i < PX; ++i) A[j][i] = 0.0f;
  float Zu[PX] = {0.f, 0.f, 0.f}, Zm[PX] = {0.f, 0.f, 0.f}, Zd[PX] = {0.f, 0.f, 0.f};  // last-level rows g-1, g, g+1

  // FOTS contact statistics of this wave's pixels
  // (count and row sum are wave-uniform per row: scalar popcounts of the mask; the column sum needs only a per-pixel
  //  counter per lane - sum_col = sum_i xg[i] * f_cpx[i] at the end)
  float f_zmax = -INFINITY;
  int f_cnt = 0, f_sr = 0;
  int f_cpx[PX] = {0, 0, 0};
  // policy observation: vertical partial sums of the <= 3 observation rows in flight
  float OA[kStreamObsActive][PX * 3];
#pragma unroll
  for (int k = 0; k < kStreamObsActive; ++k)
#pragma unroll
    for (int j = 0; j < PX * 3; ++j) OA[k][j] = 0.0f;
  const int seg_oa = do_obs ? a.obs_seg_oa[seg] : 0;
  const int seg_ob = do_obs ? a.obs_seg_ob[seg] : -1;
  int cur_o0 = seg_oa;
  float* const obs_blk = do_obs ? a.obs_part + (size_t)wv * (a.obs_nrows * a.obs_ncols * 3) : nullptr;

  auto row_of = [&](int y) -> int { return min(max(reflect_idx(y, H), 0), H - 1); };

  // horizontal reduction of one finished (or segment-final) observation row: staged through the wave's LDS row, every
  // (column, channel) cell reduced by one lane over a fixed-length window (weights 0 outside the filter / the strip)
  auto obs_flush = [&](const float (&acc)[PX * 3], int o) {
    if (o < seg_oa || o > seg_ob) return;
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < PX * 3; ++j) obs_row[lane * (PX * 3) + j] = acc[j];
    wave_lds_fence();
    for (int base = 0; base < nq * 3; base += 64) {
      const int j = base + lane;
      if (j < nq * 3) {
        const int qi = j / 3, ch = j - qi * 3;
        const float* sp = obs_row + obs_xb[qi] * 3 + ch;
        const v4f* wp = reinterpret_cast<const v4f*>(obs_wl + qi * kxp);
        float s = 0.0f;
#pragma unroll 3
        for (int t = 0; t < kxp; t += 4) {
          const v4f w4 = wp[t >> 2];
          s = fmaf(w4.x, sp[t * 3], s);
          s = fmaf(w4.y, sp[t * 3 + 3], s);
          s = fmaf(w4.z, sp[t * 3 + 6], s);
          s = fmaf(w4.w, sp[t * 3 + 9], s);
        }
        obs_blk[((o - seg_oa) * a.obs_ncols + qi) * 3 + ch] = s;
      }
    }
    wave_lds_fence();
  };

  // one finished frame row: polynomial of every pixel's table record (TT:250-255) + background + clip (TT:257-258), the RGB
  // store and the row's share of the policy observation
  // LATE STORES (round 5, -DTACEX_STREAM_LATE_STORE): gfx9 stores count in vmcnt and the counter retires in order, so the one wait of
  // an iteration - for the table / background fetches issued at its top - also waited for the RGB stores of the iteration before,
  // issued just ahead of them (knock-out build without the stores: -152 of 752 us, r04 section 9).  With this switch the RGB of the row
  // shaded in iteration y stays in registers across the loop edge and is stored in iteration y + 1 AFTER that iteration's fetches have
  // been issued; the wait becomes vmcnt(PX) - everything but those PX stores - so a store has a whole iteration to itself.
#ifdef TACEX_STREAM_LATE_STORE
  constexpr bool LATE = ROLE == kStreamFused;
#else
  constexpr bool LATE = false;
#endif
  float held[PX * 3] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int held_row = -1;       // frame row whose RGB is in held[] (-1: none)
  bool held_out = false;   // (wave-uniform) the PX stores of a held row were issued after this iteration's fetches
  auto store_held = [&]() {
    held_out = false;
    if constexpr (LATE) {
      if (held_row >= 0) {
#pragma unroll
        for (int i = 0; i < PX; ++i)
          if (valid[i])
            *reinterpret_cast<v3f*>(reinterpret_cast<char*>(a.sh.rgb + fo * 3) + ((unsigned)held_row * (unsigned)W * 12u + xc[i])) =
                (v3f){held[3 * i], held[3 * i + 1], held[3 * i + 2]};
        held_row = -1;
        held_out = true;
      }
    }
  };
  auto emit_row = [&](int e, const StreamRowInfo& ri, const v3f (&bq)[PX], const StreamRec (&pc)[PX], bool hold = false) {
    const float Y = ri.fy;
    float rgb[PX * 3];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      float p0, p1, p2;
      stream_poly(X[i], Y, pc[i], p0, p1, p2);
      rgb[3 * i + 0] = __builtin_amdgcn_fmed3f(p0 + bq[i].x, 0.0f, 1.0f);  // TT:257-258
      rgb[3 * i + 1] = __builtin_amdgcn_fmed3f(p1 + bq[i].y, 0.0f, 1.0f);
      rgb[3 * i + 2] = __builtin_amdgcn_fmed3f(p2 + bq[i].z, 0.0f, 1.0f);
      // (Three LINEAR stores per row - lane L of store j writing pixel 64 j + L after a transposition through the staging row -
      //  were built and measured in round 3: a third of the L2 write requests (44.4 M of 21 bytes per 1024 frames here, the
      //  write-through L1 forwarding every 12-byte piece on its own), the same kernel time; profiles/r03_experiments.md.)
#ifdef TACEX_DBG_NO_STORE
      if (valid[i] && rgb[3 * i] == 12345.0f)
#else
      if (valid[i] && !(LATE && hold))
#endif
        *reinterpret_cast<v3f*>(reinterpret_cast<char*>(a.sh.rgb + fo * 3) + ((unsigned)e * (unsigned)W * 12u + xc[i])) =
            (v3f){rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    }
    if constexpr (LATE) {
      if (hold) {
#pragma unroll
        for (int j = 0; j < PX * 3; ++j) held[j] = rgb[j];
        held_row = e;
      }
    }
    if (do_obs) {
      // (An `if` instead of this loop - at most one observation row retires per frame row - spares 26 register copies per row
      //  in the ISA and measured 3.3 % SLOWER, 813 vs 787 us alternating on one box: left as it is.)
      while (cur_o0 < ri.o0) {  // the oldest observation row in flight got its last frame row: reduce it horizontally
        obs_flush(OA[0], cur_o0);
#pragma unroll
        for (int j = 0; j < PX * 3; ++j) { OA[0][j] = OA[1][j]; OA[1][j] = OA[2][j]; OA[2][j] = 0.0f; }
        ++cur_o0;
      }
#pragma unroll
      for (int j = 0; j < PX * 3; ++j) {
        OA[0][j] = fmaf(ri.w0, rgb[j], OA[0][j]);
        OA[1][j] = fmaf(ri.w1, rgb[j], OA[1][j]);
        OA[2][j] = fmaf(ri.w2, rgb[j], OA[2][j]);
      }
    }
  };

  // Iteration y: input row y enters level 0; the last level leaves row zr = y - SUMR.  With shading, the row shaded in an
  // iteration is gs = y - SUMR - 2, whose neighbours (Zu, Zm, Zd) = rows gs-1, gs, gs+1 are complete at the START of the
  // iteration: its bins are computed and its background loads issued early, the levels run under their latency.
  const int ys = SHADE ? r0 - SUMR - 1 : r0 - SUMR;
  const int ye = SHADE ? r1 + SUMR + 1 : r1 - 1 + SUMR;
  float zc[PX], hc[PX] = {0.f, 0.f, 0.f};
  // ---- waves no contact can reach: no levels, no bins -------------------------------------------------------------------
  // zin is non-zero only within ext_grow rows of the frame's contact rows (zero-band skipping of the band levels, same
  // argument), S < 0 only on them: if none of the rows this wave reads (its segment + the warm-up rows) comes that close, every
  // level of every row is exactly zero, every gradient is zero and every pixel falls into the one table record of the flat
  // bin pair (TT:494-499: direction 0 where the gradient vanishes).  What remains is the shading of a flat gel: polynomial of
  // that record + background + store + observation.  A frame without contact (most frames of a manipulation episode) costs a
  // third of a frame with one.
  bool flat = false;
  if constexpr (ROLE == kStreamFused && GZ) {
    if (a.rows_ext != nullptr) {
      const int lo = a.rows_ext[4 * frame] - a.ext_grow, hi = a.rows_ext[4 * frame + 1] + a.ext_grow;
      flat = (r0 - SUMR - 1 > hi) || (r1 + SUMR + 1 < lo);
    }
  }
  // rows [lo, hi) of this wave's strip as FLAT gel: polynomial of the flat bin pair's record + background + store + observation; FOTS
  // marker taps of rows [tlo, thi): deformed gel 0, no contact (registered taps are written whether or not the contact-statistics
  // partials are: a stale tap of an earlier frame must not survive)
  auto flat_rows = [&](int lo, int hi, int tlo, int thi) {
    if constexpr (ROLE == kStreamFused && GZ) {
      if (LEVELS && (a.pix_z != nullptr || a.pix_m != nullptr) && thi > tlo) {
        const int e0 = a.rows[tlo * kStreamRowInts + 5], e1 = a.rows[(thi - 1) * kStreamRowInts + 6];
        for (int e = e0 + lane; e < e1; e += 64) {
          const int mx = a.mk_x[e];
          if (mx >= vx0 && mx < vx1) {
            const size_t o = (size_t)frame * a.n_markers + a.mk_id[e];
            if (a.pix_z != nullptr) a.pix_z[o] = 0.0f;
            if (a.pix_m != nullptr) a.pix_m[o] = 0;
          }
        }
      }
      if (hi <= lo) return;
      if (a.flat_rgb != nullptr) {
        // the flat gel's RGB is a per-context constant image: a flat row is three loads, three stores and its share of the observation
        auto load_flat = [&](int row, v3f (&q)[PX]) {
#pragma unroll
          for (int i = 0; i < PX; ++i)
            q[i] = *reinterpret_cast<const v3f*>(reinterpret_cast<const char*>(a.flat_rgb) + ((unsigned)row * (unsigned)W * 12u + xc[i]));
        };
        v3f fq[PX], fn[PX] = {(v3f)(0.0f), (v3f)(0.0f), (v3f)(0.0f)};
        load_flat(lo, fq);
        for (int e = lo; e < hi; ++e) {
          if (e + 1 < hi) load_flat(e + 1, fn);
          const StreamRowInfo ri = *reinterpret_cast<const StreamRowInfo*>(a.rows + e * kStreamRowInts);
          float rgb[PX * 3];
#pragma unroll
          for (int i = 0; i < PX; ++i) {
            rgb[3 * i] = fq[i].x; rgb[3 * i + 1] = fq[i].y; rgb[3 * i + 2] = fq[i].z;
            if (valid[i])
              *reinterpret_cast<v3f*>(reinterpret_cast<char*>(a.sh.rgb + fo * 3) + ((unsigned)e * (unsigned)W * 12u + xc[i])) = fq[i];
          }
          if (do_obs) {
            while (cur_o0 < ri.o0) {
              obs_flush(OA[0], cur_o0);
#pragma unroll
              for (int j = 0; j < PX * 3; ++j) { OA[0][j] = OA[1][j]; OA[1][j] = OA[2][j]; OA[2][j] = 0.0f; }
              ++cur_o0;
            }
#pragma unroll
            for (int j = 0; j < PX * 3; ++j) {
              OA[0][j] = fmaf(ri.w0, rgb[j], OA[0][j]);
              OA[1][j] = fmaf(ri.w1, rgb[j], OA[1][j]);
              OA[2][j] = fmaf(ri.w2, rgb[j], OA[2][j]);
            }
          }
          fq[0] = fn[0]; fq[1] = fn[1]; fq[2] = fn[2];
        }
        return;
      }
      const int code = shade_dir_bin(a.sh, 0.0f, 0.0f, 0.0f);  // magnitude bin 0
      StreamRec pcf[PX];
      {
        const StreamRec rf = stream_rec_load(code < nb_lds ? polyL + code * kStreamPolyPitch : a.sh.poly + (size_t)code * 24);
#pragma unroll
        for (int i = 0; i < PX; ++i) pcf[i] = rf;
      }
      v3f bq[PX], bn[PX] = {(v3f)(0.0f), (v3f)(0.0f), (v3f)(0.0f)};
      load_bg(lo, bq);
      for (int e = lo; e < hi; ++e) {
        if (e + 1 < hi) load_bg(e + 1, bn);
        const StreamRowInfo ri = *reinterpret_cast<const StreamRowInfo*>(a.rows + e * kStreamRowInts);
        emit_row(e, ri, bq, pcf);
        bq[0] = bn[0]; bq[1] = bn[1]; bq[2] = bn[2];
      }
    }
  };
  // ROW TRIMMING (round 5; -DTACEX_STREAM_NO_TRIM restores the full march): the last level is exactly zero farther than SUMR rows from the band levels' non-zero range
  // [z_lo, z_hi] of the frame, so of its rows [r0, r1) this wave only MARCHES over [ra, rb) = the rows within SUMR + 1 of that range; the
  // rows above and below are flat gel (loops before / after the march), and the march starts at the first non-zero input row with
  // the pipeline in its zero state (partial sums 0, S ring +inf = no restore) instead of SUMR + 1 rows of warm-up above the segment.
  int ra = r0, rb = r1, y_first = ys, y_last = ye;
  bool trimmed = false;
#ifndef TACEX_STREAM_NO_TRIM
  if constexpr (ROLE == kStreamFused && GZ) {
    if (a.rows_ext != nullptr && !flat) {
      int aa = max(r0, z_lo - SUMR - 1), bb = min(r1, z_hi + SUMR + 2);
      if (aa <= 1) aa = r0;        // row 0 takes the bins of row 1 (replicated border): emitted by the march whenever row 1 is
      if (bb >= H - 1) bb = r1;    // likewise H - 1 and H - 2
      if (bb > aa) {
        if (z_lo > SUMR + 1) y_first = max(ys, z_lo);  // (near the top border the reflected rows above it may be non-zero: full warm-up)
        y_last = min(ye, bb + SUMR + 1);
        ra = aa; rb = bb;
        trimmed = ra > r0 || rb < r1 || y_first > ys || y_last < ye;
      }
    }
  }
#endif
  if constexpr (ROLE == kStreamFused && GZ) {
    if (flat) {
      if (do_fots) f_zmax = (valid[0] || valid[1] || valid[2]) ? 0.0f : -INFINITY;
      flat_rows(r0, r1, r0, r1);
    } else if (trimmed) {
      // the taps of every marker row the march will not write (it overwrites the ones it visits; the wait below orders the two)
      if (do_fots) f_zmax = (valid[0] || valid[1] || valid[2]) ? 0.0f : -INFINITY;
      flat_rows(r0, ra, r0, r1);
    }
  }
  if constexpr (NL > 1) {
    if (!flat && y_first > ys) {  // S ring: +inf (no restore) for the rows the trimmed march never wrote
      for (int k = 0; k < NRING; ++k) {
        if constexpr (C::ring_packed()) ring3[k * 64 + lane] = StreamS3{INFINITY, INFINITY, INFINITY};
        else ring[k * 64 + lane] = (v4f){INFINITY, INFINITY, INFINITY, 0.0f};
      }
      wave_lds_fence();
    }
  }
  if (!flat) {
    issue_row(row_of(y_first), y_first & 1);
    issue_row(row_of(y_first + 1), (y_first + 1) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // Row scalars, fetched one iteration ahead with a VECTOR load (lanes 0-7: record of the row shaded, 8-15: of the row
  // entering, 16-23: of the row leaving the last level) and moved to scalar registers with v_readlane: scalar-memory loads
  // share the LDS wait counter and return out of order, so every ring read would also wait for them.
  // Lanes 24-43 / 44-63 of the same load bring the packed marker slots of the row entering / the row leaving the last level:
  // the FOTS taps of a marker row then cost three ds_bpermute and one store instead of a dependent load -> wait -> store chain
  // per marker (11 chains per marker row were ~8 % of the kernel).
  const int* rows_i = a.rows;
  auto load_info = [&](int yy) -> int {
    const int k = lane >> 3;
    const int r = row_of(k == 0 ? yy - SUMR - 2 : ((k == 1 || (lane >= 24 && lane < 24 + kStreamMkSlots)) ? yy : yy - SUMR));
    const int off = lane < 24 ? (lane & 7) : 8 + (lane - 24) % kStreamMkSlots;
    return rows_i[r * kStreamRowInts + off];
  };
  // steady iterations: every lane's record row is (y + 1) + a per-lane constant, inside the image -> offset from a scalar row base
  const int info_off = ((lane >> 3) == 0 ? -(SUMR + 2) : (((lane >> 3) == 1 || (lane >= 24 && lane < 24 + kStreamMkSlots)) ? 0 : -SUMR)) * kStreamRowInts +
                       (lane < 24 ? (lane & 7) : 8 + (lane - 24) % kStreamMkSlots);
  int cinfo = 0;  // the vector the current iteration's row scalars were unpacked from (marker slots in lanes 24..63)
  StreamRowInfo ri_g{}, ri_y{}, ri_z{};
  auto unpack_info = [&](int info) {
    if constexpr (SHADE) {
      ri_g.fy = __int_as_float(__builtin_amdgcn_readlane(info, 0)); ri_g.o0 = __builtin_amdgcn_readlane(info, 1);
      ri_g.w0 = __int_as_float(__builtin_amdgcn_readlane(info, 2)); ri_g.w1 = __int_as_float(__builtin_amdgcn_readlane(info, 3));
      ri_g.w2 = __int_as_float(__builtin_amdgcn_readlane(info, 4));
    }
    if constexpr (LEVELS) {
      ri_y.mk0 = __builtin_amdgcn_readlane(info, 8 + 5); ri_y.mk1 = __builtin_amdgcn_readlane(info, 8 + 6);
      ri_z.mk0 = __builtin_amdgcn_readlane(info, 16 + 5); ri_z.mk1 = __builtin_amdgcn_readlane(info, 16 + 6);
    }
  };
  if (!flat) {
    cinfo = load_info(y_first);
    unpack_info(cinfo);
  }
#ifdef TACEX_STREAM_CLOCK
  float clk_acc[4] = {0.f, 0.f, 0.f, 0.f};
#endif
#ifdef TACEX_STREAM_CLOCK8  // finer probe: eight sections per shaded iteration (each tick drains the LDS queue: s_memtime returns through it)
  float clk8[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  long long tk_prev = 0;
  bool tk_on = false;
#define TACEX_TICK(k) do { const long long tk_now = __builtin_readcyclecounter(); if (tk_on) clk8[k] += (float)(tk_now - tk_prev); tk_prev = tk_now; } while (0)
#else
#define TACEX_TICK(k) do { } while (0)
#endif
  int ring_slot = ((y_first % NRING) + NRING) % NRING;  // ring slot of row y (wave-uniform, advanced once per iteration)
  // One iteration of the pipeline.  ST (compile time) = a STEADY iteration: the row entering (y .. y + 2), the row leaving the last
  // level and the row shaded all lie inside the image and inside this wave's segment, away from the replicated border rows - no
  // reflection of row indices, no "does this iteration shade / emit two rows" tests, the row-scalar fetch at a per-lane constant
  // offset from a scalar base.  87 % of the iterations of a full-height item; the generic form runs the warm-up and the drain.
#ifdef TACEX_STREAM_STEADY
  auto iteration = [&](const int y, auto st_c) {
    constexpr bool ST = decltype(st_c)::value;
#else
  for (int y = flat ? y_last + 1 : y_first; y <= y_last; ++y) {
    constexpr bool ST = false;
#endif
    // ---- row y out of the ring (it landed before the previous iteration's mid_point returned); next iteration's row scalars ----
#ifdef TACEX_STREAM_CLOCK
    const long long ck0 = __builtin_readcyclecounter();
#endif
#ifdef TACEX_STREAM_CLOCK8
    tk_on = SHADE && (y - SUMR - 2) >= max(r0, 1) && (y - SUMR - 2) <= min(r1 - 1, H - 2);
    tk_prev = __builtin_readcyclecounter();
    if (tk_on) clk8[8] += 1.0f;
#endif
    read_row(y & 1, ST ? y : row_of(y), zc, hc);
    int ninfo = ST ? (rows_i + (y + 1) * kStreamRowInts)[info_off] : load_info(y + 1);
    asm volatile("" : "+v"(zc[0]), "+v"(zc[1]), "+v"(zc[2]));
    TACEX_TICK(0);  // row read-back
    // The ONE point of the iteration where this wave waits for memory: every plain load of the iteration (row scalars,
    // background, table gather) has been consumed by the caller, row y+1 (issued one iteration ago) is forced to have landed, and
    // row y+2 is issued into the slot row y was read from.  Only stores follow, so they are a full iteration old at the next
    // vmcnt(0) (on gfx9 stores count in vmcnt too, and hipcc waits vmcnt(0) at any plain load result while an LDS load is in flight).
#ifdef TACEX_STREAM_CLOCK
    long long ck1 = 0, ck2 = 0;
#endif
    auto mid_point = [&]() {
      asm volatile("" : "+v"(ninfo));
      if (LATE && held_out) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // (PX = 3 stores of the held row may stay in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TACEX_STREAM_CLOCK
      ck2 = __builtin_readcyclecounter();
#endif
      TACEX_TICK(5);  // the memory wait
      issue_row(ST ? y + 2 : row_of(y + 2), y & 1);
    };

    // ---- shading, part 1 (runs between level 0 and level 1, see below): bins of row gs, background loads in flight.
    //      Replicate padding of the gradient maps (TT:501-502): rows 0 / H-1 take the gradient of rows 1 / H-2 and are emitted
    //      together with them; columns 0 / W-1 take the bins of columns 1 / W-2. ----
    const int gs = y - SUMR - 2;
    const bool shade_now = SHADE && (ST || (gs >= max(ra, 1) && gs <= min(rb - 1, H - 2)));
    int cc[PX] = {0, 0, 0};  // table record (bin pair) of every pixel of row gs
    StreamRec pc[PX];        // its 18 polynomial coefficients (TT:250-255), fetched at the top of the iteration
    auto fetch_table = [&]() {
      // Table records: magnitude bin 0 (code < nb) from the workgroup's LDS copy - every lane reads (clamped index) - then the
      // lanes of contact pixels overwrite theirs with a gather from the table in L2; row segments without such a lane (most
      // of them) issue no vector-memory instruction for the table at all.  (Compacting the contact pixels of a row into one
      // dense gather - five instructions for up to 64 pixels instead of five per pixel slot, the polynomial values handed
      // back through LDS - was built and measured in round 3: equal within noise, profiles/r03_experiments.md; not kept.)
      bool hi[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        hi[i] = valid[i] && cc[i] >= nb_lds;  // halo lanes hold garbage bins: they must not trigger the L2 gather
#ifdef TACEX_DBG_NO_GATHER
        hi[i] = false;
#endif
        pc[i] = stream_rec_load(polyL + (hi[i] ? 0 : cc[i]) * kStreamPolyPitch);
      }
      if (__builtin_amdgcn_ballot_w64(hi[0] || hi[1] || hi[2]) != 0) {
#pragma unroll
        for (int i = 0; i < PX; ++i)
          if (hi[i]) pc[i] = stream_rec_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.sh.poly) + (unsigned)cc[i] * 96u));
      }
    };
    auto shade_part1 = [&]() {
      if constexpr (SHADE) {
        if (!shade_now) return;
        const float zl = dpp_from_left(Zm[PX - 1]), zrg = dpp_from_right(Zm[0]);
        int code[PX];
        float gx[PX], gy[PX], t2[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) {  // as shade_bins (TT:475-499), split so that the magnitude part can be skipped
          gx[i] = (Zu[i] - Zd[i]) * a.sh.gsy;
          gy[i] = ((i == 0 ? zl : Zm[i - 1]) - (i == PX - 1 ? zrg : Zm[i + 1])) * a.sh.gsx;
          t2[i] = fmaf(gx[i], gx[i], gy[i] * gy[i]);
        }
        // Direction bins (shade_dir_bin: direction 0 where sqrt(t2) == 0, TT:494-499).  ONE wave-uniform test for the row segment
        // - where the gel is flat (most rows of most frames) no arc tangent runs at all - instead of one exec-mask branch per
        // pixel slot: three branch sequences per iteration were a measurable part of the 1 700 cycles this block took.
        const int flat_code = min((int)(3.14159274101257324f * a.sh.inv_y_binr), a.sh.nb - 1);
        if (__builtin_amdgcn_ballot_w64(!(fmaxf(fmaxf(t2[0], t2[1]), t2[2]) < 1.17549435e-38f)) != 0) {
#pragma unroll
          for (int i = 0; i < PX; ++i) {
            const float dir = atan2_fast(gx[i], gy[i]);  // (NaN for a zero gradient: discarded by the select)
            const int c = min((int)((dir + 3.14159274101257324f) * a.sh.inv_y_binr), a.sh.nb - 1);
            code[i] = !(t2[i] < 1.17549435e-38f) ? c : flat_code;
          }
        } else {
#pragma unroll
          for (int i = 0; i < PX; ++i) code[i] = flat_code;
        }
        // Most row segments lie outside the contact: every |gradient| is below the first magnitude-bin edge, with a margin
        // (2e-5 relative = 2.5e-7 rad) wider than the error of the arc-tangent polynomial, so bin 0 is what the full
        // evaluation returns and the sqrt / rcp / polynomial (a third of the bin arithmetic) is skipped for the whole wave.
        if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(t2[0], t2[1]), t2[2]) >= mag0_t2) != 0) {
#pragma unroll
          for (int i = 0; i < PX; ++i) code[i] += shade_mag_bin(a.sh, t2[i]) * a.sh.nb;
        }
        // columns 0 / W-1 take the bins of columns 1 / W-2 (TT:501-502): one pixel of the strip at the image's left edge (lane
        // HL, slot 0) and one of the strip at its right edge (lane / slot of column W-1), patched under wave-uniform tests
#pragma unroll
        for (int i = 0; i < PX; ++i) cc[i] = code[i];
        if (vx0 == 0) cc[0] = lane == HL ? code[1] : code[0];
        if (vx1 == W) {
          if (edge_slot == 0) {
            const int cl = dpp_from_left_i(code[PX - 1]);
            cc[0] = lane == edge_lane ? cl : cc[0];
          } else if (edge_slot == 1) {
            cc[1] = lane == edge_lane ? code[0] : cc[1];
          } else {
            cc[2] = lane == edge_lane ? code[1] : cc[2];
          }
        }
#ifndef TACEX_STREAM_LATE_TABLE  // issued here, ahead of the levels: the gather's L2 round trip (18 % of the iteration when waited for
        fetch_table();            // on the spot) hides behind them; -2.3 % on the kernel for six more live registers
#endif
        load_bg(gs, bgq);
      }
    };
    shade_part1();
    store_held();  // (LATE) the previous iteration's RGB, behind this iteration's fetches
    TACEX_TICK(1);  // bins, table / background fetch issue

    float cur[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) cur[i] = zc[i];
    if constexpr (LEVELS) {
      // ---- S of this row into the ring; contact statistics of the rows this wave owns ----
      float S[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) S[i] = (hc[i] - sa) - sb;  // TT:441
      v4f Sring[NL > 1 ? NL - 1 : 1];
      if constexpr (NL > 1) {
        // S of the rows the restores of this iteration need (written delay(l) iterations ago): all ring reads issued up front,
        // then this row's S takes the slot of the oldest one (LDS accesses of one wave execute in program order)
        static_for<0, NL - 1>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          int sl = ring_slot - C::delay(l);
          sl += sl < 0 ? NRING : 0;
          if constexpr (C::ring_packed()) {
            const StreamS3 q = ring3[sl * 64 + lane];
            Sring[l] = (v4f){q.x, q.y, q.z, 0.0f};
          } else {
            Sring[l] = ring[sl * 64 + lane];
          }
        });
        if constexpr (C::ring_packed()) ring3[ring_slot * 64 + lane] = StreamS3{S[0], S[1], S[2]};
        else ring[ring_slot * 64 + lane] = (v4f){S[0], S[1], S[2], 0.0f};
        ring_slot = ring_slot + 1 == NRING ? 0 : ring_slot + 1;
      }
#ifdef TACEX_STREAM_NO_CONTACT_STATS  // KNOCK-OUT probe (round 6, profiles/r06_experiments.md): the contact count / row sum / column counters are
      // not accumulated (FOTS then sees no contact: timing only) and the mask is evaluated on marker rows alone - the upper bound of what
      // moving the contact statistics into another kernel could take off the tail
      if (do_fots && (ST || y >= r0) && y < r1 && a.pix_m != nullptr && ri_y.mk1 > ri_y.mk0) {
#else
      if (do_fots && (ST || y >= r0) && y < r1) {
#endif
        float gl[PX] = {0.f, 0.f, 0.f};
        if constexpr (!GZ) {
          const unsigned ro = (unsigned)row_of(y) * (unsigned)W;
#pragma unroll
          for (int i = 0; i < PX; ++i) gl[i] = a.gel[ro + xo[i]];
        }
        int mrow[PX];
        int row_cnt = 0;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          bool m;
          if constexpr (GZ) {
            m = valid[i] && S[i] < thr_eff;
          } else {
            const float J = fmin_raw(S[i], gl[i]);
            m = valid[i] && ((J - gl[i]) < thr) && (S[i] < 0.0f);  // TT:457-461
          }
          mrow[i] = m ? 1 : 0;
#ifndef TACEX_STREAM_NO_CONTACT_STATS
          f_cpx[i] += m ? 1 : 0;
          row_cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(m));
#endif
        }
        f_cnt += row_cnt; f_sr += row_cnt * y;
        if (a.pix_m != nullptr && a.mk_vec) {  // contact mask at the FOTS marker pixels of this row
          if (ri_y.mk1 > ri_y.mk0) {
            const unsigned mk = (lane >= 24 && lane < 24 + kStreamMkSlots) ? (unsigned)cinfo : 0xffffffffu;
            const int mx = (int)(mk & 0xffffu), rel = mx - cx0, owner = rel / PX, d = rel - owner * PX;
            const int v0 = __builtin_amdgcn_ds_bpermute(owner << 2, mrow[0]), v1 = __builtin_amdgcn_ds_bpermute(owner << 2, mrow[1]),
                      v2 = __builtin_amdgcn_ds_bpermute(owner << 2, mrow[2]);
            if (mk != 0xffffffffu && mx >= vx0 && mx < vx1)
              a.pix_m[(size_t)frame * a.n_markers + (mk >> 16)] = (uint8_t)(d == 0 ? v0 : (d == 1 ? v1 : v2));
          }
        } else if (a.pix_m != nullptr) {
          for (int e = ri_y.mk0; e < ri_y.mk1; ++e) {
            const int mx = a.mk_x[e];
            const int d = mx - cx0 - lane * PX;
            if (d >= 0 && d < PX && mx >= vx0 && mx < vx1)
              a.pix_m[(size_t)frame * a.n_markers + a.mk_id[e]] = (uint8_t)(d == 0 ? mrow[0] : (d == 1 ? mrow[1] : mrow[2]));
          }
        }
      }
      TACEX_TICK(2);  // S, restore ring, contact statistics, marker mask taps
      // ---- the levels: horizontal pass over the lanes, vertical scatter into the partial sums, masked restore ----
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
            out[i] = fma_to(w[WO + K - 1], h[i], A[AO + K - 2][i]);
            static_for<0, K - 2>([&](auto jc) {
              constexpr int j = K - 2 - decltype(jc)::value;  // K-2 .. 1
              A[AO + j][i] = fma_to(w[WO + j], h[i], A[AO + j - 1][i]);
            });
            A[AO][i] = w[WO] * h[i];
          }
        }
        if constexpr (l < NL - 1) {  // TT:467 Z[M] = J[M]; the final blur (TT:468-471) has no restore
          const v4f Sv = Sring[l];
          float gl[PX] = {0.f, 0.f, 0.f};
          if constexpr (!GZ) {
            const unsigned ro = (unsigned)row_of(y - C::delay(l)) * (unsigned)W;
#pragma unroll
            for (int i = 0; i < PX; ++i) gl[i] = a.gel[ro + xo[i]];
          }
#pragma unroll
          for (int i = 0; i < PX; ++i) {
            const float Si = Sv[i];
            if constexpr (GZ) {
              out[i] = Si < thr_eff ? Si : out[i];
            } else {
              const float J = fmin_raw(Si, gl[i]);
              out[i] = (((J - gl[i]) < thr) && (Si < 0.0f)) ? J : out[i];
            }
          }
        }
#pragma unroll
        for (int i = 0; i < PX; ++i) cur[i] = out[i];
      });
      TACEX_TICK(3);  // levels
      // ---- cur = last-level row zr = y - SUMR ----
      if constexpr (!SHADE) mid_point();
      const int zr = y - SUMR;
      if ((ST || zr >= r0) && zr < r1) {
        if (do_fots) {
#pragma unroll
          for (int i = 0; i < PX; ++i) f_zmax = valid[i] ? fmaxf(f_zmax, cur[i]) : f_zmax;
        }
        if (a.pix_z != nullptr && a.mk_vec) {
          if (ri_z.mk1 > ri_z.mk0) {
            const unsigned mk = lane >= 24 + kStreamMkSlots ? (unsigned)cinfo : 0xffffffffu;
            const int mx = (int)(mk & 0xffffu), rel = mx - cx0, owner = rel / PX, d = rel - owner * PX;
            const int v0 = __builtin_amdgcn_ds_bpermute(owner << 2, __float_as_int(cur[0])),
                      v1 = __builtin_amdgcn_ds_bpermute(owner << 2, __float_as_int(cur[1])),
                      v2 = __builtin_amdgcn_ds_bpermute(owner << 2, __float_as_int(cur[2]));
            if (mk != 0xffffffffu && mx >= vx0 && mx < vx1)
              a.pix_z[(size_t)frame * a.n_markers + (mk >> 16)] = __int_as_float(d == 0 ? v0 : (d == 1 ? v1 : v2));
          }
        } else if (a.pix_z != nullptr) {
          for (int e = ri_z.mk0; e < ri_z.mk1; ++e) {
            const int mx = a.mk_x[e];
            const int d = mx - cx0 - lane * PX;
            if (d >= 0 && d < PX && mx >= vx0 && mx < vx1)
              a.pix_z[(size_t)frame * a.n_markers + a.mk_id[e]] = d == 0 ? cur[0] : (d == 1 ? cur[1] : cur[2]);
          }
        }
        if constexpr (!SHADE) {  // levels role: the last level goes to HBM (12 contiguous bytes per lane)
          if (valid[0] && valid[PX - 1]) {
            *reinterpret_cast<v3f*>(a.z_out + fo + (size_t)zr * W + xg[0]) = (v3f){cur[0], cur[1], cur[2]};
          } else {
#pragma unroll
            for (int i = 0; i < PX; ++i)
              if (valid[i]) a.z_out[fo + (size_t)zr * W + xg[i]] = cur[i];
          }
        }
      }
    }
    TACEX_TICK(4);  // last-level taps (deformed gel at the marker pixels, maximum)
    // ---- shading, part 2: table records, polynomial, background, clip, store, observation ----
    if constexpr (SHADE) {
      if (shade_now) {
#ifdef TACEX_STREAM_LATE_TABLE  // A/B probe: table fetch on the spot
        fetch_table();
#endif
#ifdef TACEX_STREAM_CLOCK
        ck1 = __builtin_readcyclecounter();
#endif
        // results of the plain loads are taken HERE (see mid_point)
        asm volatile("" : "+v"(pc[0].c0), "+v"(pc[0].c1), "+v"(pc[0].c2), "+v"(pc[0].c3), "+v"(pc[0].c4), "+v"(pc[1].c0), "+v"(pc[1].c1),
                     "+v"(pc[1].c2), "+v"(pc[1].c3), "+v"(pc[1].c4));
        asm volatile("" : "+v"(pc[2].c0), "+v"(pc[2].c1), "+v"(pc[2].c2), "+v"(pc[2].c3), "+v"(pc[2].c4), "+v"(bgq[0]), "+v"(bgq[1]),
                     "+v"(bgq[2]));
      }
      mid_point();
      if constexpr (ST) {
        emit_row(gs, ri_g, bgq, pc);
      } else if (shade_now) {
        const int e_lo = gs == 1 ? 0 : gs, e_hi = gs == H - 2 ? H - 1 : gs;  // rows emitted by this iteration (ascending)
        for (int e = e_lo; e <= e_hi; ++e) {
          if (e < ra || e >= rb) continue;
          StreamRowInfo ri = ri_g;
          v3f bq[PX] = {bgq[0], bgq[1], bgq[2]};
          if (e != gs) {  // a replicated border row: its own feature / background / observation row (twice per frame)
            ri = *reinterpret_cast<const StreamRowInfo*>(a.rows + e * kStreamRowInts);
            load_bg(e, bq);
          }
          emit_row(e, ri, bq, pc, /*hold=*/e_lo == e_hi);
        }
      }
    }
    TACEX_TICK(6);  // row issue, polynomial, stores, observation
    cinfo = ninfo;
    unpack_info(ninfo);
#pragma unroll
    for (int i = 0; i < PX; ++i) { Zu[i] = Zm[i]; Zm[i] = Zd[i]; Zd[i] = cur[i]; }
    TACEX_TICK(7);  // row scalars of the next iteration
#ifdef TACEX_STREAM_CLOCK
    {
      const long long ck3 = __builtin_readcyclecounter();
      if (ck1 != 0) { clk_acc[0] += (float)(ck1 - ck0); clk_acc[1] += (float)(ck2 - ck1); clk_acc[2] += (float)(ck3 - ck2); clk_acc[3] += 1.0f; }
    }
#endif
#ifdef TACEX_STREAM_STEADY
  };
  {
    // BUILT AND MEASURED SLOWER (round 5, profiles/r05_experiments.md section 2): the steady loop is 1155 instructions against 1306 and holds
    // 41 SGPR-spill reloads against 84, but with three loop bodies in the function the allocator spills 85 SGPRs (64 before), 20 of them
    // beyond the spill VGPR into scratch (80 B/lane): 871 us per 1024 frames against 805.  Kept behind the macro as the A/B partner.
    // steady range: gs = y - SUMR - 2 in [max(r0, 2), min(r1 - 1, H - 3)] (shades exactly one row), y + 2 <= H - 1 (no reflection);
    // its lower end implies y >= r0, zr >= r0 and non-negative rows in the row-scalar fetch
    int y = flat ? ye + 1 : ys;
    const int st_lo = SHADE ? max(r0, 2) + SUMR + 2 : ye + 1, st_hi = SHADE ? min(min(r1 - 1, H - 3) + SUMR + 2, H - 3) : ye;
    for (; y <= ye && y < st_lo; ++y) iteration(y, std::false_type{});
    for (; y <= st_hi; ++y) iteration(y, std::true_type{});
    for (; y <= ye; ++y) iteration(y, std::false_type{});
  }
#else
  }
#endif
#ifdef TACEX_STREAM_CLOCK8
  if (lane == 0 && SHADE) {
    float* dbg = a.sh.rgb + fo * 3 + rem * 9;
    for (int k = 0; k < 9; ++k) dbg[k] = clk8[k];
  }
#endif
#ifdef TACEX_STREAM_CLOCK
  if (lane == 0 && SHADE) {  // probe output over the first pixels of the frame: [rem][compute before wait, wait, after wait, shaded rows]
    float* dbg = a.sh.rgb + fo * 3 + rem * 4;
    dbg[0] = clk_acc[0]; dbg[1] = clk_acc[1]; dbg[2] = clk_acc[2]; dbg[3] = clk_acc[3];
  }
#endif
  store_held();  // (LATE) the last shaded row
  if constexpr (ROLE == kStreamFused && GZ) {
    if (trimmed && rb < r1) flat_rows(rb, r1, 0, 0);  // the flat rows below the march (ascending row order: the observation rows retire in order)
  }
  if constexpr (SHADE) {
    if (do_obs) {  // observation rows still in flight at the end of the segment (another segment adds its share)
      obs_flush(OA[0], cur_o0);
      obs_flush(OA[1], cur_o0 + 1);
      obs_flush(OA[2], cur_o0 + 2);
      // rows of the block this segment never reached stay unwritten: the finishing kernel only reads [seg_oa, seg_ob]
    }
  }
  if (do_fots) {  // one record per wave: no atomics; fots_combine_kernel adds the records of an env
    f_zmax = wave_scan_max_lane63(f_zmax);
    int f_sc = 0;
#pragma unroll
    for (int i = 0; i < PX; ++i) f_sc += f_cpx[i] * xg[i];
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

// RGB of the undeformed gel (every gradient zero: direction 0 -> the flat bin pair's record, TT:494-499): the same arithmetic, in the same
// order, as emit_row - what a flat row of the streaming tail would compute for every frame is a per-context constant image
__global__ __launch_bounds__(256) void stream_flat_image_kernel(ShadeArgs sh, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= sh.W) return;
  const int code = shade_dir_bin(sh, 0.0f, 0.0f, 0.0f);
  const StreamRec r = stream_rec_load(sh.poly + (size_t)code * 24);
  float p0, p1, p2;
  stream_poly(sh.fx[x], sh.fy[y], r, p0, p1, p2);
  const float* bg = sh.bg + ((size_t)y * sh.W + x) * 3;
  float* o = out + ((size_t)y * sh.W + x) * 3;
  o[0] = __builtin_amdgcn_fmed3f(p0 + bg[0], 0.0f, 1.0f);
  o[1] = __builtin_amdgcn_fmed3f(p1 + bg[1], 0.0f, 1.0f);
  o[2] = __builtin_amdgcn_fmed3f(p2 + bg[2], 0.0f, 1.0f);
}

// Items of a launch sorted by weight, heaviest first: key = rows of the item's segment that lie on (or within a few rows of) the
// frame's contact rows - there the shading gathers table records and runs the magnitude arc tangent - 0 for a flat item (no
// contact within the pyramid's reach: no levels, no bins).  Counting sort by one workgroup; the order inside a bucket is
// whatever the atomics give (every item's output is independent of where it runs).  (Keying on the clock ticks each item took
// in the previous launch of the same shard was built and measured: WORSE than this geometric key - 752 vs 720 us per 1024
// frames - the measured time of an item contains the contention it ran under; profiles/r03_experiments.md.)
__global__ __launch_bounds__(1024) void stream_order_kernel(const int* __restrict__ rows_ext, int n_items, int per_frame, int nseg,
                                                            int seg_rows, int H, int reach, int* __restrict__ order) {
  constexpr int kKeys = 1024;
  __shared__ int hist[kKeys], start[kKeys];
  for (int k = threadIdx.x; k < kKeys; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  auto key_of = [&](int item) {
    const int frame = item / per_frame, seg = (item - frame * per_frame) % nseg;
    const int r0 = seg * seg_rows, r1 = min(H, r0 + seg_rows);
    const int lo = rows_ext[4 * frame], hi = rows_ext[4 * frame + 1];
    if ((r0 - reach > hi) || (r1 + reach < lo)) return 0;  // flat (frames without contact: lo = H, hi = -1)
    const int pad = 4;
    const int c = min(r1 - 1, hi + pad) - max(r0, lo - pad) + 1;
    return min(kKeys - 1, 1 + max(c, 0));
  };
  for (int i = threadIdx.x; i < n_items; i += blockDim.x) atomicAdd(&hist[key_of(i)], 1);
  __syncthreads();
  if (threadIdx.x == 0) {  // keys never exceed 1 + seg_rows + 2 pad: the serial scan walks those alone (it was most of this kernel's 17 us)
    int acc = 0;
    for (int k = min(kKeys - 1, seg_rows + 9); k >= 0; --k) { start[k] = acc; acc += hist[k]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_items; i += blockDim.x) order[atomicAdd(&start[key_of(i)], 1)] = i;
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
template <int ROLE, int... KS>
static bool stream_geometry_t(int W, int* nstrips, int* strip_w) {
  using C = StreamCfg<ROLE, KS...>;
  int ns = (W + C::VW - 1) / C::VW;
  int sw = (((W + ns - 1) / ns) + 3) & ~3;
  if (sw > C::VW) { ++ns; sw = (((W + ns - 1) / ns) + 3) & ~3; }
  *nstrips = ns;
  *strip_w = sw;
  return sw <= C::VW;
}

// fused level sets with a streaming instantiation (the same sets the tiled tail covers)
static int stream_variant(int n_fused, int k0) {
  if (n_fused == 4 && k0 == 9) return 0;   // <9,5,3,5>  320x240
  if (n_fused == 3 && k0 == 9) return 1;   // <9,5,9>    640x480 (k = 15 stays a band level)
  if (n_fused == 3 && k0 == 5) return 2;   // <5,3,5>    320x240 with k = 9 as a band level (three waves per SIMD)
  return -1;
}

static bool stream_split() {
  static const int v = getenv("TACEX_STREAM_SPLIT") ? atoi(getenv("TACEX_STREAM_SPLIT")) : 0;
  return v != 0;
}

bool stream_supported(int n_fused, int k0, int H, int W) {
  static const int en = getenv("TACEX_TAIL_STREAM") ? atoi(getenv("TACEX_TAIL_STREAM")) : 1;
  return en != 0 && stream_variant(n_fused, k0) >= 0 && H >= 16 && W >= 16 && W % 4 == 0;
}

// geometry of the kernel that SHADES (strip layout of the RGB / observation partial sums) and of the one that runs the LEVELS
// (strip layout of the FOTS partial records); identical in fused mode
bool stream_geometry(int n_fused, int k0, int W, int* nstrips, int* strip_w, int* lv_nstrips, int* lv_strip_w) {
  const int v = stream_variant(n_fused, k0);
  if (v < 0) return false;
  if (stream_split()) {
    if (!stream_geometry_t<kStreamShade>(W, nstrips, strip_w)) return false;
    return v == 0 ? stream_geometry_t<kStreamLevels, 9, 5, 3, 5>(W, lv_nstrips, lv_strip_w)
         : v == 1 ? stream_geometry_t<kStreamLevels, 9, 5, 9>(W, lv_nstrips, lv_strip_w)
                  : stream_geometry_t<kStreamLevels, 5, 3, 5>(W, lv_nstrips, lv_strip_w);
  }
  const bool ok = v == 0 ? stream_geometry_t<kStreamFused, 9, 5, 3, 5>(W, nstrips, strip_w)
                : v == 1 ? stream_geometry_t<kStreamFused, 9, 5, 9>(W, nstrips, strip_w)
                         : stream_geometry_t<kStreamFused, 5, 3, 5>(W, nstrips, strip_w);
  *lv_nstrips = *nstrips; *lv_strip_w = *strip_w;
  return ok;
}

// vertical segments per strip: enough waves to give every SIMD `per_simd` of them, at most kStreamMaxSeg; a segment costs
// `warm` extra rows (the pipeline's warm-up), so it should be several times that long
int stream_segments(int B, int nstrips, int H, int warm, int per_simd) {
  static const int forced = getenv("TACEX_STREAM_SEGS") ? atoi(getenv("TACEX_STREAM_SEGS")) : 0;
  int nseg = forced > 0 ? forced : (1024 * per_simd + B * nstrips - 1) / (B * nstrips);
  if (nseg < 1) nseg = 1;
  if (nseg > kStreamMaxSeg) nseg = kStreamMaxSeg;
  const int min_rows = forced > 0 ? 24 : (3 * warm > 24 ? 3 * warm : 24);
  while (nseg > 1 && (H / nseg < min_rows || H - (nseg - 1) * ((H + nseg - 1) / nseg) < 4)) --nseg;
  return nseg;
}
// waves per SIMD the fused kernel of this level set is compiled for (sizes the segment count: a launch should bring a whole
// number of resident rounds)
int stream_waves_per_simd(int n_fused, int k0) { return stream_variant(n_fused, k0) == 2 ? TACEX_STREAM3_WAVES : 2; }
int stream_warm_rows(int n_fused, int k0, bool levels_kernel) {
  const int v = stream_variant(n_fused, k0);
  const int sum_r = v == 0 ? 9 : (v == 1 ? 10 : 5);
  if (!stream_split()) return 2 * sum_r + 2;
  return levels_kernel ? 2 * sum_r : 2;
}

template <bool GZ, int ROLE, int... KS>
static hipError_t launch_stream_k(const StreamArgs& a, hipStream_t st) {
  using C = StreamCfg<ROLE, KS...>;
  const int waves = a.B * a.nstrips * a.nseg;
  const int nwg = (waves + kStreamWaves - 1) / kStreamWaves;
  const dim3 grid(nwg);
  static const size_t lds_pad = getenv("TACEX_STREAM_LDS_PAD") ? (size_t)atoi(getenv("TACEX_STREAM_LDS_PAD")) : 0;  // occupancy A/B hook
  const size_t lds = C::lds_bytes() + lds_pad;
  auto kern = taxim_stream_kernel<GZ, ROLE, C::min_waves(), KS...>;
  static size_t granted[64] = {};  // per kernel instantiation and device
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, granted); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(64 * kStreamWaves), lds, st, a);
  return hipGetLastError();
}

template <int ROLE, int... KS>
static hipError_t launch_stream(const StreamArgs& a, bool gel_zero, hipStream_t st) {
  return gel_zero ? launch_stream_k<true, ROLE, KS...>(a, st) : launch_stream_k<false, ROLE, KS...>(a, st);
}

static void fill_shade_args(ShadeArgs& sh, const ShadeParams* sp, float* rgb, int B, int H, int W) {
  sh.poly = sp->poly_dev; sh.bg = sp->bg_nhwc_dev; sh.fx = sp->fx_dev; sh.fy = sp->fy_dev; sh.rgb = rgb;
  sh.idx_out = nullptr; sh.H = H; sh.W = W; sh.B = B; sh.nb = sp->nb; sh.pixmm = sp->pixmm;
  sh.calib_h = (float)sp->calib_h; sh.calib_w = (float)sp->calib_w; sh.x_binr = sp->x_binr; sh.y_binr = sp->y_binr;
  sh.gsy = (float)(0.5 * H / sp->calib_h / (double)sp->pixmm); sh.gsx = (float)(0.5 * W / sp->calib_w / (double)sp->pixmm);
  sh.inv_x_binr = (float)(1.0 / (double)sp->x_binr); sh.inv_y_binr = (float)(1.0 / (double)sp->y_binr);
}

hipError_t run_stream_flat_image(const ShadeParams* sp, int H, int W, hipStream_t st) {
  if (!sp->flat_rgb_dev) return hipErrorInvalidValue;
  ShadeArgs sh{};
  fill_shade_args(sh, sp, nullptr, 1, H, W);
  hipLaunchKernelGGL(stream_flat_image_kernel, dim3((W + 255) / 256, H), dim3(256), 0, st, sh, sp->flat_rgb_dev);
  return hipGetLastError();
}

// TACEX_STREAM_ORDER=0: items in frame order (A/B path); the split kernels always run that way
static bool stream_sorted() {
  static const int sorted = getenv("TACEX_STREAM_ORDER") ? atoi(getenv("TACEX_STREAM_ORDER")) : 1;
  return sorted != 0;
}

// The item order of a fused streaming-tail launch, as a launch of its own: it depends on the contact rows alone, so the pipeline issues it
// beside the band levels (as soon as the last depth pass has left the rows) instead of between the levels' join and the tail.
// Returns hipSuccess and sets *launched when the order buffer will hold the launch's order.
hipError_t run_stream_order(const LevelDesc* lv, int n_levels, int n_fused, const StreamPlan& plan, int B, int H, const int* rows_ext, int ext_grow,
                            int* order_buf, hipStream_t st, bool* launched) {
  *launched = false;
  if (stream_split() || !stream_sorted() || !order_buf || !lv[0].gel_zero || !rows_ext) return hipSuccess;
  const int v = stream_variant(n_fused, lv[n_levels - n_fused].kw);
  const int n_items = B * plan.nstrips * plan.nseg;
  const int sum_r = v == 0 ? 9 : (v == 1 ? 10 : 5);
  hipLaunchKernelGGL(stream_order_kernel, dim3(1), dim3(1024), 0, st, rows_ext, n_items, plan.nstrips * plan.nseg, plan.nseg, plan.seg_rows, H,
                     ext_grow + sum_r + 1, order_buf);
  const hipError_t e = hipGetLastError();
  *launched = e == hipSuccess;
  return e;
}

hipError_t run_stream_tail(const LevelDesc* lv, int n_levels, int n_fused, const float* zin, const float* hm, const float* gel,
                           const float* sa, const float* sb, const float* pd, const ShadeParams* sp, float* rgb, float* z_last,
                           int B, int H, int W, float contact_scale, const StreamPlan& plan, float* obs_part,
                           FotsReduce* fots_part, int fots_stride, float* pix_z, uint8_t* pix_m, hipStream_t st,
                           const int* rows_ext, int ext_grow, int* order_buf, bool order_done) {
  const bool sorted = stream_sorted();
  StreamArgs a{};
  a.rows_ext = lv[0].gel_zero ? rows_ext : nullptr; a.ext_grow = ext_grow;
  a.zin = zin; a.hm = hm; a.gel = lv[0].gel_zero ? nullptr : gel; a.shift_a = sa; a.shift_b = sb; a.pdepth = pd;
  a.H = H; a.W = W; a.B = B; a.contact_scale = contact_scale;
  for (int i = 0; i < n_fused; ++i) a.taps[i] = lv[n_levels - n_fused + i].taps_w_dev;
  fill_shade_args(a.sh, sp, rgb, B, H, W);
#ifndef TACEX_STREAM_NO_TRIM
  a.flat_rgb = sp->flat_rgb_dev;
#endif
  {
    const double t1 = tan((double)sp->x_binr) * (1.0 - 2e-5);
    a.mag0_t2 = (float)(t1 * t1);
  }
  a.rows = static_cast<const int*>(plan.rows); a.mk_vec = plan.mk_vec ? 1 : 0;
  StreamArgs sh = a;  // arguments of the kernel that shades
  sh.nstrips = plan.nstrips; sh.strip_w = plan.strip_w; sh.nseg = plan.nseg; sh.seg_rows = plan.seg_rows;
  if (obs_part && plan.obs_ready) {
    sh.obs_part = obs_part;
    sh.obs_kxp = plan.obs_kxp;
    sh.obs_xlo = plan.obs.xlo; sh.obs_xcnt = plan.obs.xcnt; sh.obs_wx = plan.obs.wx; sh.obs_kx = plan.obs.kx;
    sh.obs_strip_q0 = plan.obs_strip_q0; sh.obs_strip_nq = plan.obs_strip_nq;
    sh.obs_seg_oa = plan.obs_seg_oa; sh.obs_seg_ob = plan.obs_seg_ob;
    sh.obs_nrows = plan.obs_nrows; sh.obs_ncols = plan.obs_ncols;
  }
  StreamArgs lvl = a;  // ... and of the one that runs the levels (the same kernel in fused mode)
  lvl.nstrips = plan.lv_nstrips; lvl.strip_w = plan.lv_strip_w; lvl.nseg = plan.lv_nseg; lvl.seg_rows = plan.lv_seg_rows;
  StreamArgs& f = stream_split() ? lvl : sh;  // who writes the FOTS by-products
  f.fots_part = fots_part; f.fots_stride = fots_stride;
  if (pix_z && pix_m && plan.mk_x) {
    f.pix_z = pix_z; f.pix_m = pix_m; f.n_markers = plan.n_markers;
    f.mk_x = plan.mk_x; f.mk_id = plan.mk_id;
  }
  const int v = stream_variant(n_fused, lv[n_levels - n_fused].kw);
  const bool gz = lv[0].gel_zero;
  if (!stream_split()) {
    if (order_done) {
      sh.order = order_buf;  // (run_stream_order has been issued for this launch)
    } else if (sorted && order_buf && gz && rows_ext) {  // heaviest items first (see the kernel's "work distribution" note)
      const int n_items = B * sh.nstrips * sh.nseg;
      const int sum_r = v == 0 ? 9 : (v == 1 ? 10 : 5);
      hipLaunchKernelGGL(stream_order_kernel, dim3(1), dim3(1024), 0, st, rows_ext, n_items, sh.nstrips * sh.nseg, sh.nseg, sh.seg_rows, H,
                         ext_grow + sum_r + 1, order_buf);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      sh.order = order_buf;
    }
    if (v == 0) return launch_stream<kStreamFused, 9, 5, 3, 5>(sh, gz, st);
    if (v == 1) return launch_stream<kStreamFused, 9, 5, 9>(sh, gz, st);
    if (v == 2) return launch_stream<kStreamFused, 5, 3, 5>(sh, gz, st);
    return hipErrorInvalidValue;
  }
  if (!z_last) return hipErrorInvalidValue;
  lvl.z_out = z_last;
  hipError_t e = v == 0 ? launch_stream<kStreamLevels, 9, 5, 3, 5>(lvl, gz, st)
               : v == 1 ? launch_stream<kStreamLevels, 9, 5, 9>(lvl, gz, st) : launch_stream<kStreamLevels, 5, 3, 5>(lvl, gz, st);
  if (e != hipSuccess) return e;
  sh.zin = z_last;
  return launch_stream<kStreamShade>(sh, true, st);
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

#ifdef TACEX_STREAM_PROBE_DEEP
// VERDICT r04 item 6 ("cut one HBM round trip of Z or show the build that fails"): the last BAND level (k = 17 at 320x240, k = 15 at
// 640x480) moved into the streaming tail's level chain would save its 8 B/px write + re-read.  These instantiations are that build -
// `hipcc -DTACEX_STREAM_PROBE_DEEP -Rpass-analysis=kernel-resource-usage` (profiles/r05_experiments.md section 4): 256 VGPRs + 61 / 52
// spilled (ScratchSize 160 / 152 B per lane) against 242 / 245 and none; a 15-row restore ring and 34 partial sums per pixel make
// 105 KB of LDS per workgroup (one workgroup per CU instead of two: +25 % by TACEX_STREAM_LDS_PAD, r03 section 1); the halo grows to
// 18 columns per side (three strips of 108 columns instead of two of 160 at 320x240: 33 % instead of 15 % redundant columns) and the
// level arithmetic from 44 to 78 FMAs per pixel on a kernel that is VALU-issue-bound.  Not wired into the runtime.
template __global__ void taxim_stream_kernel<true, kStreamFused, 2, 17, 9, 5, 3, 5>(StreamArgs);
template __global__ void taxim_stream_kernel<true, kStreamFused, 2, 15, 9, 5, 9>(StreamArgs);
static_assert(StreamCfg<kStreamFused, 17, 9, 5, 3, 5>::lds_bytes() > 80 * 1024, "deep variant: one workgroup per CU");
#endif

int stream_obs_lds_floats() { return kStreamObsLdsFloats; }
int stream_obs_max_cols() { return kStreamObsMaxCols; }

}  // namespace tacex
