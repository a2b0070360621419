// Internal declarations shared by the HIP translation units of libtacex_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tacex {

struct LevelDesc {
  int kw = 1, kh = 1;
  bool gel_zero = false;   // the context's gel map is identically 0 (GelSight Mini): kernels skip its loads
  bool same_taps = false;  // taps_w == taps_h element-wise (sigma_w == sigma_h): tuned band kernel eligible
  float* taps_w_dev = nullptr;
  float* taps_h_dev = nullptr;
  float* taps_mfma_dev = nullptr; // per-lane band weights for blur_mfma_kernel: [V | H][(16 + 2 RA)/4 k-steps][64 lanes]
  float* taps_pad_dev = nullptr;  // zero-padded taps for the looped band kernel: [16 zeros | K taps | 48 zeros]
};

// contact statistics FOTS needs per env (fots_marker_sim.py:130-141): max of the deformed gel, contact-mask pixel count and
// row / column index sums.  The fused tail writes one per WAVE of every tile (no barrier, no atomics); see fots_kernels.hip.
struct FotsReduce {
  float zmax;
  int count;
  int sum_row;  // <= 480 * 307200 fits int32
  int sum_col;
};
constexpr int kTailWavesPerTile = 8;
constexpr int kTailMaxMarkersPerTile = 16;
// FOTS marker pixels per 64 x 32 tile of the fused tail (device tables, built by tacex_taxim_set_fots_taps)
struct FotsTaps {
  const int* mk_tile = nullptr;  // [tiles_per_frame][kTailMaxMarkersPerTile]: marker << 16 | y_in_tile << 8 | x_in_tile
  const int* mk_cnt = nullptr;   // [tiles_per_frame]
  int n_markers = 0;
};

struct ShadeParams {
  int H = 0, W = 0, nb = 0, calib_h = 0, calib_w = 0;
  float pixmm = 0.f, x_binr = 0.f, y_binr = 0.f;
  float* poly_dev = nullptr;     // (nb, nb, 24)
  float* bg_nhwc_dev = nullptr;  // (H, W, 3)
  float* fx_dev = nullptr;       // (W,)
  float* fy_dev = nullptr;       // (H,)
  float* flat_rgb_dev = nullptr; // (H, W, 3) RGB of the undeformed gel: clip(poly[flat bin pair](x, y) + background), run_stream_flat_image
};

hipError_t run_frame_min(const float* in, bool from_depth, float* hm_out, float* fmin, float* indent,
                         uint8_t* cam_u8, int B, int npix, float near_mm, float far_m, float far_mm, float gelpad_h,
                         float gelpad_dmin, hipStream_t st);
hipError_t run_indenter_height_map(const float* desc, float* hm, float* fmin, float* indent, int B, int H, int W, float pixmm,
                                   float gel_top_mm, float far_clip_mm, float gelpad_h, float gelpad_dmin, hipStream_t st);
hipError_t run_press_depth(const float* fmin, const float* press, float* sa, float* sb, float* pd, int B,
                           int no_shift, hipStream_t st);
bool blur_level_single_kernel(const LevelDesc& lv, bool first, int H, int W);  // one launch, no use of the shared scratch image
hipError_t run_blur_level(const LevelDesc& lv, const float* src, const float* hm, const float* gel,
                          const float* sa, const float* sb, const float* pd, float* dst, float* tmp,
                          uint8_t* mask_out, int B, int H, int W, float contact_scale, int restore,
                          bool first, hipStream_t st, const int* rows_ext = nullptr, int ext_grow = 0, int ext_grow_x = 0);
bool frame_rows_supported(int H, int W);
hipError_t run_fill_rows(int* rows, int B, int H, int W, hipStream_t st);
hipError_t run_frame_rows(const float* in, bool from_depth, float* hm_out, float* fmin, float* indent, uint8_t* cam_u8,
                          const float* press_in, int* rows_out, int B, int H, int W, float near_mm, float far_m, float far_mm,
                          float gelpad_h, float gelpad_dmin, hipStream_t st);
hipError_t run_shade(const ShadeParams& sp, const float* z, float* rgb, uint8_t* idx_out, int B,
                     hipStream_t st);
hipError_t run_resize_aa(const float* src, int sh, int sw, float* dst, int dh, int dw, int B, int C, float* tmp,
                         hipStream_t st);

struct ShadowParams {  // shadow branch tables (taxim_shadow.hip), set by tacex_taxim_set_shadow
  bool ready = false;
  int ndir = 0, nfan = 0, nheight = 0, nstep = 0;
  int wl = 0, wr = 0, wt = 0, wb = 0;
  float depth0 = 0.4f, height_prec = 0.1f, disc_prec = 0.1f, step_x = 0.f, step_y = 0.f;
  float* fan_dev = nullptr;     // (ndir, nfan)
  float* fan_cos_dev = nullptr; // (ndir, nfan) host-computed float32 cos / sin of the fan angles
  float* fan_sin_dev = nullptr;
  float* table_dev = nullptr;   // (ndir, nheight, nstep, 4)
  int sblur_kw = 1, sblur_kh = 1, final_kw = 1, final_kh = 1;
  float* sblur_taps_w_dev = nullptr; float* sblur_taps_h_dev = nullptr;
  float* final_taps_w_dev = nullptr; float* final_taps_h_dev = nullptr;
};
hipError_t run_shadow_rays(const ShadowParams& sw, const ShadeParams& sp, const float* z, const uint8_t* mask, const float* gel,
                           const float* gdir, float* shadow_min, int B, hipStream_t st);
hipError_t run_shadow(const ShadowParams& sw, const ShadeParams& sp, const float* z, const uint8_t* mask, const float* gel,
                      float* rgb, float* ws_raw, float* ws_shadow, float* ws_gdir, float* ws_tmp, int B, hipStream_t st);

// H-pass contraction map of blur_mfma_kernel: window column that k-step ks of lane group g (= lane >> 4) contracts.  The window
// (KS = WIN / 4 k-steps per group, WIN / 4 = KS chunks of four columns) is dealt out in chunk PAIRS (c, c + 2): groups 0 / 1 take
// the two chunks of the first KS / 4 pairs, groups 2 / 3 those of the rest.  A ds_read_b128 is served in four groups of 16 lanes
// - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (guides/MI355X_MICROARCH.md, LDS) - i.e. eight rows of lane group
// g and the OTHER eight rows of group g + 1: with rows 0-3 / 12-15 on even and rows 4-11 on odd tile rows (mfma_h_row) and an odd
// pitch / 4, the two halves fall on 16-byte slots of different parity when their chunks differ by 2: no bank conflict (the
// linear map u = KS g + ks had 5 of 16 slots hit twice: LDS conflict ratio 0.42 at k = 61).
__host__ __device__ inline int mfma_h_window_col(int KS, int g, int ks) {
  const int p = (ks >> 2) + (g >> 1) * (KS >> 2);  // chunk pair
  return 4 * (4 * (p >> 1) + (p & 1) + 2 * (g & 1)) + (ks & 3);
}
// tile row that lane i (= lane & 15) of the H-pass reads, restores and stores
__host__ __device__ inline int mfma_h_row(int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) + 1 : 2 * (i < 4 ? i : i - 8); }

// fused tail (taxim_tail.hip): trailing small-kernel levels + shading in one LDS-tiled kernel
int tail_levels(const LevelDesc* lv, int n_levels, int H, int W);
// triangle-filter tables of the antialiased policy-observation down-sample (one entry per output row / column):
// first source index, tap count, the un-normalised taps (stride ky / kx) and their sum - built once per (oh, ow)
struct ObsTables {
  const int* ylo; const int* ycnt; const float* ysum; const float* wy; int ky;
  const int* xlo; const int* xcnt; const float* xsum; const float* wx; int kx;
  int oh, ow;
  int ymax, xmax;  // longest filter (taps) per axis
};
hipError_t run_tail(const LevelDesc* lv, int n_levels, int n_fused, const float* zin, const float* hm, const float* gel,
                    const float* sa, const float* sb, const float* pd, float* z_out, uint8_t* mask_out,
                    const ShadeParams* sp, float* rgb, float* obs_part, const ObsTables* obs_tab, FotsReduce* fots_part,
                    const FotsTaps* taps, float* pix_z, uint8_t* pix_m, int B, int H, int W, float contact_scale, hipStream_t st);
size_t tail_tiles_per_frame(int H, int W);
hipError_t run_obs_finish(const float* part, void* obs, bool u8, const ObsTables& t, int H, int W, int B, int nry, int ncx,
                          hipStream_t st);
bool tail_obs_geom(int n_fused, int k0, int* nry, int* ncx, int* ky, int* kx);
hipError_t run_obs_to_u8(const float* src, uint8_t* dst, size_t n, hipStream_t st);
size_t obs_part_floats(int H, int W, int B, int nry, int ncx);
bool obs_fusable(const ObsTables& t, int H, int W, int n_fused, int k0);  // geometry the fused tail reduction is compiled for

// streaming fused tail (taxim_stream.hip): wave-autonomous line-buffer pipeline, the default where an instantiation exists
constexpr int kStreamMaxLevels = 5;
constexpr int kStreamMaxSeg = 8;
// per-frame-row scalars the streaming kernel needs, packed so that ONE 32-byte scalar load per row fetches them (issued an
// iteration ahead): polynomial feature y, the observation rows the row feeds (first row + 3 weights), its FOTS marker range
struct StreamRowInfo {
  float fy;
  int o0;
  float w0, w1, w2;
  int mk0, mk1;  // markers mk_x / mk_id [mk0, mk1) lie in this row
  int pad;
};
// the per-row table holds kStreamRowInts ints per frame row: the StreamRowInfo record (8) followed by the row's FOTS markers as
// packed slots (column | marker index << 16, 0xffffffff = empty), so that the ONE vector load that fetches the row scalars of
// an iteration also brings the markers of the two rows the iteration taps (lanes 24..43 / 44..63)
constexpr int kStreamRowInts = 32;
constexpr int kStreamMkSlots = 20;
struct StreamPlan {
  int nstrips = 0, strip_w = 0, nseg = 0, seg_rows = 0;              // kernel that shades (RGB, observation partial sums)
  int lv_nstrips = 0, lv_strip_w = 0, lv_nseg = 0, lv_seg_rows = 0;  // kernel that runs the levels (FOTS partial records)
  // policy observation of one (oh, ow, nseg): device tables, valid when obs_ready
  bool obs_ready = false;
  ObsTables obs{};
  const void* rows = nullptr;  // (H,) StreamRowInfo: feature y, observation rows + weights, marker range of every frame row
  int obs_kxp = 0;             // column-filter window length (taps padded to a multiple of 4)
  const int* obs_strip_q0 = nullptr; const int* obs_strip_nq = nullptr;
  const int* obs_seg_oa = nullptr; const int* obs_seg_ob = nullptr;
  int obs_nrows = 0, obs_ncols = 0;
  // FOTS marker pixels as a CSR over frame rows (device), nullptr when no taps are set
  const int* mk_x = nullptr; const int* mk_id = nullptr; int n_markers = 0;
  bool mk_vec = false;  // every row's markers fit the packed slots of the row table (else the kernel walks the CSR)
};
int stream_obs_lds_floats();
int stream_obs_max_cols();
bool stream_supported(int n_fused, int k0, int H, int W);
bool stream_geometry(int n_fused, int k0, int W, int* nstrips, int* strip_w, int* lv_nstrips, int* lv_strip_w);
int stream_segments(int B, int nstrips, int H, int warm_rows, int waves_per_simd);
int stream_warm_rows(int n_fused, int k0, bool levels_kernel);
int stream_waves_per_simd(int n_fused, int k0);
// z_last: (B,H,W) scratch for the last level (split mode: the levels kernel writes it, the shading kernel reads it)
hipError_t run_stream_flat_image(const ShadeParams* sp, int H, int W, hipStream_t st);  // fills sp->flat_rgb_dev
hipError_t run_stream_tail(const LevelDesc* lv, int n_levels, int n_fused, const float* zin, const float* hm, const float* gel,
                           const float* sa, const float* sb, const float* pd, const ShadeParams* sp, float* rgb, float* z_last,
                           int B, int H, int W, float contact_scale, const StreamPlan& plan, float* obs_part,
                           FotsReduce* fots_part, int fots_stride, float* pix_z, uint8_t* pix_m, hipStream_t st, const int* rows_ext = nullptr, int ext_grow = 0,
                           int* order_buf = nullptr,  // (B * strips * segments) device ints for the launch's item order (nullptr: frame order)
                           bool order_done = false);  // the order has been issued by run_stream_order (the tail's stream is behind it)
hipError_t run_stream_order(const LevelDesc* lv, int n_levels, int n_fused, const StreamPlan& plan, int B, int H, const int* rows_ext, int ext_grow,
                            int* order_buf, hipStream_t st, bool* launched);
hipError_t run_obs_finish_stream(const float* part, void* obs, bool u8, const StreamPlan& plan, int B, hipStream_t st);

// More than 48 KB of dynamic LDS is an opt-in per kernel AND per device (one process may drive several GPUs): `granted` is the
// per-kernel table (a function-local static at the launch site) of the largest size already granted on every device.
inline hipError_t ensure_dynamic_lds(const void* kern, size_t lds, size_t (&granted)[64]) {
  if (lds <= 48 * 1024) return hipSuccess;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (lds <= granted[dev]) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) granted[dev] = lds;
  return e;
}

// thread-local error string (tacex_last_error)
void set_error(const char* fmt, ...);
int fail_hip(hipError_t e, const char* what);

}  // namespace tacex
