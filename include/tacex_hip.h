/*
 * tacex_hip.h - C ABI of libtacex_hip.so, the MI355X (gfx950) implementation of the TacEx tactile hot path.
 *
 * The reference (DH-Ng/TacEx) has no FFI: its hot path is PyTorch code behind the Python plugin classes
 * GelSightSimulator / GelSightSensor.  This header is the boundary the build introduces UNDER those
 * classes; each entry point names the reference code it replaces.  Short names:
 *   TT = source/tacex/tacex/simulation_approaches/gpu_taxim/sim/taxim_torch.py
 *   TS = source/tacex/tacex/simulation_approaches/gpu_taxim/taxim_sim.py
 *   GS = source/tacex/tacex/gelsight_sensor.py
 *   MM = source/tacex/tacex/simulation_approaches/fots/sim/marker_motion.py
 *   FS = source/tacex/tacex/simulation_approaches/fots/fots_marker_sim.py
 *   US/UO/UA = source/tacex_uipc/tacex_uipc/{sim/uipc_sim.py,objects/uipc_object.py,sim/uipc_attachments.py}
 *   VT = source/tacex/tacex/simulation_approaches/fem_based/sim/tactile_sensor_sapienipc_modified.py
 *
 * Conventions
 *   - plain C types only; every `*_dev` pointer is DEVICE memory owned by the caller (PyTorch, usually);
 *   - no allocation, no synchronisation inside a compute call: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream);
 *   - one context per device, not thread-safe per context;
 *   - every function returns 0 on success, non-zero on error; tacex_last_error() gives the message
 *     (the Python layer turns it into RuntimeError / ValueError like the reference's exceptions);
 *   - images are row-major float32, batch first: height maps (B,H,W) in millimetres, RGB (B,H,W,3).
 */
#ifndef TACEX_HIP_H
#define TACEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TACEX_MAX_LEVELS 8
#define TACEX_ABI_VERSION 15

typedef struct tacex_taxim_ctx tacex_taxim_ctx;
typedef struct tacex_fots_ctx tacex_fots_ctx;
typedef struct tacex_fem_ctx tacex_fem_ctx;

const char* tacex_last_error(void);
int tacex_abi_version(void);
/* Number of HIP devices visible / name of the gcn arch of `device_id` (for loud failure messages). */
int tacex_device_count(int* count);
int tacex_device_arch(int device_id, char* buf, size_t buflen);

/* ---------------------------------------------------------------------------------------------
 * Taxim optical path.  Tables are HOST pointers, copied to the device once at creation
 * (replaces TaximTorch.__init__, TT:50-130, and the lru-cached tables TT:136-164).
 * ------------------------------------------------------------------------------------------- */
typedef struct tacex_taxim_params {
  int32_t height, width;        /* tactile image size H, W (TS:49-54 tactile_img_res = (W,H)) */
  int32_t calib_height, calib_width; /* calibration size (params.json sensor.h / sensor.w: 480, 640) */
  float pixmm;                  /* params.json sensor.pixmm (0.0295) */
  int32_t num_bins;             /* params.json sensor.num_bins (125) */
  float contact_scale;          /* params.json simulator.contact_scale (0.4), TT:459 */
  int32_t n_levels;             /* pyramid levels + 1 final blur (7), TT:464-471 */
  int32_t ksize_w[TACEX_MAX_LEVELS]; /* odd kernel widths  per level, TT:396-403 */
  int32_t ksize_h[TACEX_MAX_LEVELS]; /* odd kernel heights per level */
  const float* taps_w[TACEX_MAX_LEVELS]; /* normalised float32 taps, ksize_w[l] each, TT:362-366 */
  const float* taps_h[TACEX_MAX_LEVELS];
  const float* poly;            /* (3, num_bins, num_bins, 6) float32 = stack(grad_b,grad_g,grad_r)/255, TT:73-80 */
  const float* gel_map;         /* (H, W) float32, mm, max == 0, TT:82-90,159-164 */
  const float* background;      /* (3, H, W) float32 in [0,1], TT:92-94,136-137,414-430 */
  const float* feat_x;          /* (W,) float32: linspace(0, calib_width,  W+1)[:-1], TT:139-157 */
  const float* feat_y;          /* (H,) float32: linspace(0, calib_height, H+1)[:-1] */
} tacex_taxim_params;

int tacex_taxim_create(int device_id, const tacex_taxim_params* params, tacex_taxim_ctx** out);
void tacex_taxim_destroy(tacex_taxim_ctx* ctx);

/* Bytes of caller-provided device scratch tacex_taxim_render / tacex_taxim_deform need for B frames. */
size_t tacex_taxim_workspace_bytes(const tacex_taxim_ctx* ctx, int num_frames);

/* GS:581-593 (_get_height_map) + GS:557-579 (_get_camera_depth) + TS:115-131 (compute_indentation_depth)
 * in one pass over the camera depth image:
 *   hm_mm      = (isinf(depth_m) ? far_clip_m : depth_m) * 1000                       (B,Hc,Wc)
 *   frame_min  = min over the frame of hm_mm                                           (B,)
 *   indent_mm  = d <= gelpad_height ? (gelpad_height - d) * 1000 : 0,
 *                d = max(frame_min/1000 - gelpad_to_camera_min_distance, 0)            (B,)  [nullable]
 *   cam_u8     = uint8(((hm_mm - near_clip_m*1000) / (far_clip_m*1000)) * 255)  (sic)  (B,Hc,Wc) [nullable]
 * The clip range is passed as DOUBLES: the reference multiplies the Python doubles by 1000 and torch rounds the product once
 * to float32 (GS:573-574), so cam_u8 is bit-exact.  depth_m_dev may alias hm_mm_dev.
 *   frame_rows (B,4) int32 [nullable, needs indent_mm]: [0], [1] first / last frame ROW holding a pixel below the press plane,
 *                i.e. with S = (hm - frame_min) - indent < 0 (TT:441) - (height, -1) when there is none; [2], [3] first / last
 *                such COLUMN ((width, -1); (0, width - 1) for widths the pass cannot track).  By-product of the same pass (per-row
 *                and per-column minima); the render uses it to skip pyramid bands and 64-column blocks that cannot be non-zero
 *                (tacex_taxim_set_frame_rows + TACEX_FLAG_HAVE_FRAME_ROWS).  (ABI 10: two ints per frame before.) */
int tacex_height_map_from_depth(const float* depth_m_dev, double near_clip_m, double far_clip_m,
                                float gelpad_height_m, float gelpad_to_camera_min_distance_m,
                                float* hm_mm_dev, float* frame_min_dev, float* indent_mm_dev,
                                uint8_t* cam_u8_dev, int32_t* frame_rows_dev, int num_frames, int height, int width,
                                void* stream);

/* The same pass, handed to the NEXT tacex_taxim_render / _render_obs / _deform call of `ctx` instead of being launched now (ABI 11).
 * A render whose hm_mm / frame_min / press buffers are this pass's hm_mm_dev / frame_min_dev / indent_mm_dev (same frame count,
 * TACEX_FLAG_HAVE_FRAME_MIN, a frame-rows buffer that is this pass's) runs it INSIDE the pass: per chunk of the band levels, on the
 * chunk's own stream, so that one chunk's depth pass (HBM-bound) overlaps the previous chunk's band levels (matrix pipe) - the
 * reference's sequence _get_height_map -> compute_indentation_depth -> optical_simulation (GS:229-263, 581-593) as one launch
 * sequence.  Any other render on the context, or tacex_taxim_flush_deferred, runs the pending pass in full first: the outputs are
 * complete, in stream order, no later than the end of the next call on the context.  Frame size = the context's.  At most one
 * pending pass per context (a second call fails). */
int tacex_taxim_defer_height_map_from_depth(tacex_taxim_ctx* ctx, const float* depth_m_dev, double near_clip_m, double far_clip_m,
                                            float gelpad_height_m, float gelpad_to_camera_min_distance_m,
                                            float* hm_mm_dev, float* frame_min_dev, float* indent_mm_dev,
                                            uint8_t* cam_u8_dev, int32_t* frame_rows_dev, int num_frames);
/* Runs a pending deferred pass now (no-op without one). */
int tacex_taxim_flush_deferred(tacex_taxim_ctx* ctx, void* stream);

/* Height-map SOURCE: rasterise one analytic indenter per env into the height map (mm), with the per-frame
 * minimum and the indentation depth of TS:115-131 in the same pass.  Stands in for the TiledCamera depth render
 * (GS:229-263 -> _get_height_map GS:581-593) when the contact geometry is a primitive; 4 B/px written, nothing read.
 * indenters_dev (B, 8) f32: [kind, cx_px, cy_px, r_px, angle_rad, press_mm, cx2_px, cy2_px]
 *   kind 0 sphere, 1 lying cylinder (radius r/2), 2 wedge with 45 degree flanks, 3 two spheres, < 0 no contact.
 * depth = min(gel_top_mm - press_mm + profile(x, y), far_clip_mm), profile in mm above the indenter's lowest point. */
int tacex_height_map_from_indenters(const float* indenters_dev, float pixmm, float gel_top_mm, float far_clip_mm,
                                    float gelpad_height_m, float gelpad_to_camera_min_distance_m, float* hm_mm_dev,
                                    float* frame_min_dev, float* indent_mm_dev, int num_frames, int height, int width,
                                    void* stream);

/* Height-map SOURCE for arbitrary rigid indenters: the depth image the IsaacLab TiledCamera of the reference
 * hands _get_height_map (GS:229-263, 581-593 - "distance_to_image_plane" in metres, inf where nothing is seen inside the
 * clipping range), rendered here from one shared triangle mesh and one rigid pose per env.
 *   verts_dev (V,3) f32 object frame [m]; tris_dev (T,3) int32; pos_dev (B,3) f32 translation and quat_dev (B,4) f32 rotation
 *   (wxyz, normalised by the kernel) taking object coordinates into the CAMERA frame (x right, y down, z along the optical axis);
 *   pinhole intrinsics fx, fy, cx, cy in pixels, pixel (i, j) sampled at its centre (j + 0.5, i + 0.5);
 *   bounding_sphere: HOST pointer to {cx, cy, cz, r} of a sphere around the mesh in the object frame, nullable (image tiles
 *   the sphere cannot project onto skip the triangle loop);
 *   depth_m_dev (B,H,W) f32 out.  Feed it to tacex_height_map_from_depth. */
int tacex_depth_from_mesh(const float* verts_dev, const int32_t* tris_dev, int num_verts, int num_tris,
                          const float* pos_dev, const float* quat_dev, float fx, float fy, float cx, float cy,
                          float near_clip_m, float far_clip_m, const float* bounding_sphere, float* depth_m_dev, int num_envs, int height,
                          int width, void* stream);

/* TS:115-131 on an existing mm height map. frame_min_dev (B,) is also written (re-used by the render);
 * frame_rows_dev (B,4) int32 nullable: contact row / column ranges as in tacex_height_map_from_depth. */
int tacex_indentation_depth(const float* hm_mm_dev, float gelpad_height_m,
                            float gelpad_to_camera_min_distance_m, float* frame_min_dev,
                            float* indent_mm_dev, int32_t* frame_rows_dev, int num_frames, int height, int width,
                            void* stream);

/* Shadow branch tables (TaximTorch.__init__ shadow calibration TT:96-126 + the per-shape parameters of TT:260-346).
 * Optional: only needed before a render with TACEX_FLAG_WITH_SHADOW. Host pointers, copied once. */
typedef struct tacex_shadow_params {
  int32_t num_directions;      /* 63  (shadowDirections) */
  int32_t num_fan_rays;        /* 4   = int(2 * fan_angle / fan_precision), TT:102 */
  int32_t num_heights;         /* 24 */
  int32_t num_steps;           /* 51  (longest table entry; shorter ones padded with +inf, TT:118-126) */
  const float* fan_angles;     /* (num_directions, num_fan_rays): direction + linspace(-fan_angle, fan_angle), TT:103-105 */
  const float* fan_cos;        /* (num_directions, num_fan_rays) float32 cos / sin of fan_angles as the HOST library computes them */
  const float* fan_sin;        /*   (torch.cos / torch.sin of the float32 table, TT:299,303): the device multiplies by these bits */
  const float* table;          /* (3, num_directions, num_heights, num_steps), RGB order, already / 255, +inf padded */
  int32_t win_left, win_right, win_top, win_bottom; /* composite window of the two box-dilation rounds, TT:261-272 */
  float shadow_depth_0;        /* 0.4, TT:97 */
  float height_precision;      /* params.json simulator.height_precision (0.1) */
  float discretize_precision;  /* params.json simulator.discretize_precision (0.1) */
  float step_x, step_y;        /* shadow_step(shape)[1], [0] (sic: x uses the height-scaled value), TT:298-305 */
  int32_t blur_kw, blur_kh;    /* shadow_blur_sigma kernel, TT:339-342 */
  const float* blur_taps_w;
  const float* blur_taps_h;
} tacex_shadow_params;

int tacex_taxim_set_shadow(tacex_taxim_ctx* ctx, const tacex_shadow_params* params);
/* extra scratch (beyond tacex_taxim_workspace_bytes) a render with TACEX_FLAG_WITH_SHADOW needs, placed right after it */
size_t tacex_taxim_shadow_workspace_bytes(const tacex_taxim_ctx* ctx, int num_frames);
/* The ray march of the shadow branch alone (TT:261-337): deformed gel z_dev (B,H,W) mm, shrunken contact mask mask_dev
 * (B,H,W) uint8, gradient direction grad_dir_dev (B,H,W) -> shadow_min_dev (B,H,W,3): per pixel and channel the minimum
 * table value over all ray samples that hit it, +inf where none does (what torch_scatter.scatter_min leaves in
 * `shadow_img`, TT:324-336).  Ring pixels, table bins and sample coordinates are integer work in the reference's op order
 * (float32 multiply, add, truncate - no fused multiply-add): parity tests compare this map EXACTLY. */
int tacex_taxim_shadow_rays(tacex_taxim_ctx* ctx, const float* z_dev, const uint8_t* mask_dev, const float* grad_dir_dev,
                            float* shadow_min_dev, int num_frames, void* stream);

/* Flags for tacex_taxim_render / tacex_taxim_deform */
#define TACEX_FLAG_NO_SHIFT      1u  /* press_depth=None: use the height map as is (TT:188-189 skipped) */
#define TACEX_FLAG_HAVE_FRAME_MIN 2u /* frame_min_dev already holds min(hm) per frame (skip that pass) */
#define TACEX_FLAG_OBS_U8         8u /* render_obs only: obs_out_dev is uint8 (B,obs_h,obs_w,3) = floor(255 x + 0.5) */
#define TACEX_FLAG_WITH_SHADOW    4u /* render only: shadow branch TT:260-346 (needs tacex_taxim_set_shadow + extra scratch) */
#define TACEX_FLAG_HAVE_FRAME_ROWS 16u /* with HAVE_FRAME_MIN: the buffer of tacex_taxim_set_frame_rows describes THESE height maps */

/* Contact row / column ranges of the height maps handed to the render (written by tacex_height_map_from_depth /
 * tacex_indentation_depth into a caller-owned (capacity_frames, 4) int32 buffer).  The deformed gel of TT:443-473 is exactly
 * zero on every pyramid band - and every 64-column block of a band - whose input window lies outside the ranges (zero gel map
 * only): those are stored as zeros without reading or multiplying anything.  Used only by calls that pass TACEX_FLAG_HAVE_FRAME_MIN | TACEX_FLAG_HAVE_FRAME_ROWS
 * with num_frames <= capacity_frames; without HAVE_FRAME_MIN the library computes the ranges in its own minimum pass.
 * nullptr disables. */
int tacex_taxim_set_frame_rows(tacex_taxim_ctx* ctx, const int32_t* frame_rows_dev, int capacity_frames);

/* TaximSimulator.optical_simulation (TS:80-113) -> Taxim.render_direct (TI:153-163) ->
 * TaximTorch._render_impl / __render no-shadow branch (TT:174-258), output already NHWC (TS:109-111).
 *   hm_mm_dev   (B,H,W) height map in mm (H,W = ctx size)
 *   press_dev   (B,)    press depth in mm (= indentation depth); ignored with TACEX_FLAG_NO_SHIFT
 *   frame_min_dev (B,)  scratch/in: per-frame min of hm (see TACEX_FLAG_HAVE_FRAME_MIN)
 *   rgb_dev     (B,H,W,3) float32 in [0,1]
 *   z_out_dev   (B,H,W) deformed gel (mm) after the final blur, nullable      (TT:443-473 result #1)
 *   mask_out_dev(B,H,W) uint8 shrunken contact mask, nullable                 (TT:473 result #2)
 *   workspace_dev: tacex_taxim_workspace_bytes(ctx, B) bytes of device scratch */
int tacex_taxim_render(tacex_taxim_ctx* ctx, const float* hm_mm_dev, const float* press_dev,
                       float* frame_min_dev, float* rgb_dev, float* z_out_dev, uint8_t* mask_out_dev,
                       void* workspace_dev, int num_frames, unsigned flags, void* stream);

/* FOTS contact statistics as a by-product of the render: when a buffer is set, the fused tail kernel (where one exists:
 * tacex_taxim_fots_partials_per_env > 0) stores one 16-byte partial {max Z, mask count, row sum, column sum} per wave and
 * tile while the frame is in LDS; tacex_fots_markers_partials consumes them.  Only calls with num_frames <= capacity_frames
 * write (the buffer holds capacity_frames * partials_per_env records); nullptr disables. */
int tacex_taxim_fots_partials_per_env(const tacex_taxim_ctx* ctx);
/* ... and the deformed gel / contact mask AT THE MARKER PIXELS only (FOTS looks nothing else up, MM:152-166): with the taps
 * set, the fused tail also fills z_pix_dev / mask_pix_dev (capacity_frames, n_markers) and the caller may pass
 * z_out_dev = mask_out_dev = NULL to the render - 5 B/px of stores less.  marker_x / marker_y: HOST int32 pixel positions
 * (the FOTS marker grid); markers outside the image are skipped.  NULL buffers disable. */
int tacex_taxim_set_fots_taps(tacex_taxim_ctx* ctx, const int32_t* marker_x, const int32_t* marker_y, int n_markers,
                              float* z_pix_dev, uint8_t* mask_pix_dev, int capacity_frames);
int tacex_taxim_set_fots_partials(tacex_taxim_ctx* ctx, void* partials_dev, int capacity_frames);

/* tacex_taxim_render + the low-resolution POLICY OBSERVATION in the same pass: obs_out_dev (B,obs_h,obs_w,3) is the
 * antialiased bilinear down-sample of the RGB frame (torchvision resize semantics, as tasks feed 32x32x3 to the policy:
 * tacex_tasks/.../ball_rolling_tactile_rgb.py:303,318).  Where the fused tail kernel exists the horizontal half of the
 * filter is accumulated while the frame is still in LDS (the full-resolution frame is never re-read), otherwise it falls
 * back to the two-pass resize.  obs_scratch_dev: B * (max(H * obs_w, obs_h * W) + obs_h * obs_w) * 3 floats.
 * obs_out_dev is float32, or uint8 with TACEX_FLAG_OBS_U8 (the image a CNN policy consumes; a quarter of the bytes in the
 * per-step observation all-gather). */
int tacex_taxim_render_obs(tacex_taxim_ctx* ctx, const float* hm_mm_dev, const float* press_dev, float* frame_min_dev,
                           float* rgb_dev, float* z_out_dev, uint8_t* mask_out_dev, void* workspace_dev,
                           float* obs_scratch_dev, void* obs_out_dev, int obs_h, int obs_w, int num_frames,
                           unsigned flags, void* stream);

/* __get_shifted_height_map + __compute_gel_pad_deformation only (TT:432-473), as the FOTS wrapper calls
 * them (FS:128-129). Same arguments as above without the shading. */
int tacex_taxim_deform(tacex_taxim_ctx* ctx, const float* hm_mm_dev, const float* press_dev,
                       float* frame_min_dev, float* z_out_dev, uint8_t* mask_out_dev,
                       void* workspace_dev, int num_frames, unsigned flags, void* stream);

/* Shading only: deformed gel (B,H,W) mm -> RGB (TT:237-258): normals, bins, polynomial, + background, clip.
 * idx_out_dev (B,H,W,2) uint8 [idx_mag, idx_dir] is optional (parity tests). */
int tacex_taxim_shade(tacex_taxim_ctx* ctx, const float* z_dev, float* rgb_dev, uint8_t* idx_out_dev,
                      int num_frames, void* stream);

/* torchvision resize(bilinear, antialias=True) of a batch of single-channel images, used when the camera
 * resolution differs from the tactile resolution (TS:88-89, FS:121-122). */
int tacex_resize_bilinear_aa(const float* src_dev, int src_h, int src_w, float* dst_dev, int dst_h,
                             int dst_w, int num_frames, void* stream);
/* Same filter on channels-last images (B,H,W,C): produces the low-resolution policy observation from the
 * tactile RGB frame (tasks feed 32x32x3 to the policy, tacex_tasks/.../ball_rolling_tactile_rgb.py:303,318),
 * which is what the multi-GPU observation all-gather carries. */
int tacex_resize_bilinear_aa_nhwc(const float* src_dev, int src_h, int src_w, float* dst_dev, int dst_h,
                                  int dst_w, int channels, int num_frames,
                                  float* tmp_dev /* num_frames*dst_h*src_w*channels floats: separable two-pass
                                                    (recommended for down-sampling); NULL = single pass */,
                                  void* stream);

/* Ablation / test hook: 1 (default) = trailing small-kernel levels + shading run as ONE fused kernel where a tuned
 * instantiation exists (320x240, 640x480): the wave-autonomous streaming kernel (taxim_stream.hip) for plain renders, the
 * LDS-tiled kernel (taxim_tail.hip) when the full deformed-gel / mask frames are requested; 2 = always the LDS-tiled kernel;
 * 0 = every level as its own kernel + separate shade kernel. */
int tacex_taxim_set_fused_tail(tacex_taxim_ctx* ctx, int enabled);

/* Frames one pass of the kernel sequence covers for a call with num_frames frames: large shards are walked in chunks whose
 * level buffers stay resident in the 256 MB Infinity Cache (== num_frames when the shard is rendered in one pass).
 * bench.py needs it to turn per-launch durations into bytes per launch. */
int tacex_taxim_chunk_frames(const tacex_taxim_ctx* ctx, int num_frames);

/* Optional per-stage timing with hipEvents on the launch stream (bench.py's roofline leg).
 * Stages: 0 = frame-min, 1..n_levels = blur levels, n_levels+1 = shade, n_levels+2 = fused tail. */
int tacex_taxim_set_profiling(tacex_taxim_ctx* ctx, int enabled);
/* Synchronises the recorded events; returns accumulated milliseconds and launch count, then resets. */
int tacex_taxim_read_profile(tacex_taxim_ctx* ctx, int stage, double* total_ms, int* launches);
int tacex_taxim_num_stages(const tacex_taxim_ctx* ctx);
const char* tacex_taxim_stage_name(const tacex_taxim_ctx* ctx, int stage);

/* ---------------------------------------------------------------------------------------------
 * FOTS marker-displacement field (MM:22-219 driven per env by FS:114-184).
 * ------------------------------------------------------------------------------------------- */
typedef struct tacex_fots_params {
  int32_t height, width;          /* tactile image size */
  int32_t num_markers_row, num_markers_col; /* 9, 11 (FSC:46-53) */
  const int32_t* marker_x;        /* HOST (rows*cols,) initial pixel x, row-major (row, col), MM:59-76 */
  const int32_t* marker_y;        /* HOST (rows*cols,) initial pixel y */
  double lamb[3];                 /* [dilate, shear, twist] = [0.00125, 0.00021, 0.00038], FS:77 */
  float mm2pix;                   /* 19.58, FSC:36 */
  float shear_max;                /* 10 px, MM:78 */
  float theta_max_deg;            /* 60 deg, MM:90 */
} tacex_fots_params;

int tacex_fots_create(int device_id, const tacex_fots_params* params, tacex_fots_ctx** out);
void tacex_fots_destroy(tacex_fots_ctx* ctx);

/* Per-env trajectory state (FS:101-103,168,176-177): only traj[0], traj[-1] and len(traj) are ever read
 * (MM:177-205).  Layout of traj_state_dev: (B, 8) float32 = [len, x0, y0, th0, xl, yl, thl, n_contacts of the last step].
 * Bytes needed: */
size_t tacex_fots_state_bytes(int num_envs);
size_t tacex_fots_workspace_bytes(int num_envs);

/* FS:130-182 for all envs in one go:
 *   z_dev (B,H,W) deformed gel, mask_dev (B,H,W) uint8 (both from tacex_taxim_deform / _render),
 *   indent_dev (B,) mm, theta_dev (B,) yaw of the indenter in the sensor frame (replaces the
 *   FrameTransformer read-out FS:147-159), traj_state_dev in/out,
 *   markers_dev (B,2,M,2) float32: [initial | current] x (x, y), FS:90-99. */
int tacex_fots_markers(tacex_fots_ctx* ctx, const float* z_dev, const uint8_t* mask_dev,
                       const float* indent_dev, const float* theta_dev, float* traj_state_dev,
                       float* markers_dev, void* workspace_dev, int num_envs, void* stream);

/* With pixels_compact != 0, z_dev / mask_dev are the (num_envs, num_markers) arrays tacex_taxim_set_fots_taps fills
 * instead of full (num_envs, H, W) frames. */
int tacex_fots_markers_compact(tacex_fots_ctx* ctx, const float* z_pix_dev, const uint8_t* mask_pix_dev,
                               const float* indent_dev, const float* theta_dev, float* traj_state_dev,
                               float* markers_dev, void* workspace_dev, const void* partials_dev,
                               int partials_per_env, int num_envs, void* stream);
/* Same, with the per-env contact statistics (max of the deformed gel, mask centroid sums; FS:130-141) taken from the
 * partials a tacex_taxim_render* / _deform call wrote (tacex_taxim_set_fots_partials) instead of re-reading z / mask. */
int tacex_fots_markers_partials(tacex_fots_ctx* ctx, const float* z_dev, const uint8_t* mask_dev,
                                const float* indent_dev, const float* theta_dev, float* traj_state_dev,
                                float* markers_dev, void* workspace_dev, const void* partials_dev,
                                int partials_per_env, int num_envs, void* stream);

/* Marker IMAGE and RGB x marker overlay (FS:346-384 `draw_markers`, FS:265-272) for all envs:
 *   canvas (H+24, W+24) uint8 = 255; for every marker in index order (later ones overwrite):
 *     u = x + 0.5 + 12, v = y + 0.5 + 12 (float64); patch = table[floor(frac(u) SR)][floor(frac(v) SR)][patch_w] (12 x 12);
 *     stamped at (floor(u) - 6, floor(v) - 6) when it lies fully inside the canvas;  image = canvas[12:-12, 12:-12].
 *   markers_dev (B,2,M,2) f32 as written by tacex_fots_markers (the CURRENT positions, index 1, are drawn);
 *   patch_table_dev (SR, SR, size_slots, 12, 12) uint8 - the reference draws it with cv2 (`generate_patch_array`, FS:387-446);
 *   patch_w = floor((marker_size - base_circle_radius) * SR) (FS:370-373: 15 for marker_size 3);
 *   img_dev (B,H,W) uint8 [nullable]; overlay_dev (B,H,W,3) uint8 = uint8(float64(rgb * 255) * marker / 255) [nullable,
 *   needs rgb_dev (B,H,W,3) f32]. */
int tacex_fots_marker_image(const float* markers_dev, const uint8_t* patch_table_dev, int super_resolution_ratio,
                            int size_slots, int patch_w, const float* rgb_dev, uint8_t* img_dev, uint8_t* overlay_dev,
                            int num_envs, int num_markers, int height, int width, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gelpad FEM inner step (what libuipc's world.advance() runs for the StableNeoHookean gelpad,
 * US:250-252, UO:442-470, UA:364-428).  Batched: `num_envs` independent copies of one tet mesh.
 * All floating point is float64.  Parity: UNPINNED (libuipc is an un-vendored submodule); the
 * restatement follows Smith, de Goes, Kim 2018 "Stable Neo-Hookean Flesh Simulation".
 * ------------------------------------------------------------------------------------------- */
typedef struct tacex_fem_params {
  int32_t num_verts, num_tets;
  const double* rest_positions;  /* HOST (V,3) */
  const int32_t* tets;           /* HOST (T,4) */
  double youngs, poisson;        /* Pa, - (UO:59,76-88: E = 0.01 MPa, nu = 0.49) */
  double density;                /* kg/m^3 (1e3) */
  double dt;                     /* s (US:57: 0.01) */
  double gravity[3];             /* (0,0,-9.8) US:60 */
  double constraint_strength_ratio; /* SoftPositionConstraint strength (UA:38: 100) */
} tacex_fem_params;

int tacex_fem_create(int device_id, const tacex_fem_params* params, tacex_fem_ctx** out);
void tacex_fem_destroy(tacex_fem_ctx* ctx);
size_t tacex_fem_workspace_bytes(const tacex_fem_ctx* ctx, int num_envs);

/* Per-element Stable-Neo-Hookean energy, gradient (12) and Hessian (12x12, optionally PSD-projected),
 * one tet per lane.  x_dev (B,V,3); outputs nullable: energy_dev (B,T), grad_dev (B,T,12), hess_dev (B,T,144). */
int tacex_fem_element_terms(tacex_fem_ctx* ctx, const double* x_dev, double* energy_dev, double* grad_dev,
                            double* hess_dev, int project_psd, int num_envs, void* stream);

/* Total incremental potential per env (inertia + dt^2 * elastic + soft position constraints),
 * wavefront-reduced: E_dev (B,).  x_tilde_dev (B,V,3) is the inertial target; constrained_dev (B,V) uint8
 * and aim_dev (B,V,3) describe UA:364-385's animation targets (nullable = no constraints). */
int tacex_fem_energy(tacex_fem_ctx* ctx, const double* x_dev, const double* x_tilde_dev,
                     const uint8_t* constrained_dev, const double* aim_dev, double* E_dev,
                     void* workspace_dev, int num_envs, void* stream);

/* Assembled nodal gradient (B,V,3) of the same potential - atomics-free vertex gather over incident tets. */
int tacex_fem_gradient(tacex_fem_ctx* ctx, const double* x_dev, const double* x_tilde_dev,
                       const uint8_t* constrained_dev, const double* aim_dev, double* g_dev,
                       void* workspace_dev, int num_envs, void* stream);

/* IPC contact of the gelpad SURFACE against one analytic indenter per env (first slice; reference cfg
 * US:103-124 `Contact`: constitution "ipc", d_hat 1e-3, resistance 10 GPa - libuipc's Compute Contact / Detect Candidates).
 * Every surface vertex v gets the barrier term  dt^2 * stiffness * area_v * b(d_v / d_hat),  b(s) = -(s-1)^2 ln s on (0,1)
 * (Li et al. 2020 eq. 6), d_v = signed distance to the indenter (sphere |x - c| - R, half-space n.(x - c), capsule: distance to the axis segment - R); the energy,
 * gradient, PSD-projected Hessian (b'' n n^T) enter tacex_fem_energy / _gradient / _newton_step, and the Newton step length is
 * first cut to 0.9 x the conservative bound min_v d_v / |dx_v| (the CCD filter of the line search: no vertex can cross the
 * indenter surface), then backtracked on the energy as before.  Friction and mesh-mesh contact are not part of this slice.
 *   vertex_area_host (V) HOST f64: a third of the area of the surface triangles around a vertex, 0 for interior vertices
 *                    (copied once; NULL keeps the previous table);
 *   indenters_dev (num_envs, 8) f64 [kind, cx, cy, cz, radius, nx, ny, nz]: kind 0 none, 1 sphere, 2 half-space (unit
 *                    normal n through c), 3 capsule (radius around the segment c -+ (nx, ny, nz): the vector is HALF the
 *                    axis); read by every later compute call until replaced; NULL disables contact.
 * Replaced tables are freed by the setter (it drains the device first: a set-up call, not a step-path call). */
int tacex_fem_set_contact(tacex_fem_ctx* ctx, const double* vertex_area_host, double d_hat, double stiffness,
                          const double* indenters_dev);

/* Coulomb friction of the gelpad surface against its env's indenter (US:103-124: enable_friction, default_friction_ratio,
 * eps_velocity), the IPC way (Li et al. 2020, eq. 18-20) with normal force and contact normal lagged per time step - taken at the
 * state the step starts from, the normal force capped by the contact reaction there (= the previous step's normal force); the
 * tangential sliding is measured from the positions the time step starts at and relative to the indenter's own displacement
 * since the previous tacex_fem_step (its positions are kept in the workspace), potential mu lam f0(|u|) smoothed below
 * eps_velocity * dt.  The displacement is the TRANSLATION of the indenter row (cx, cy, cz) between two steps: the caller may move
 * an indenter by mutating the rows in place or by handing a new indenters_dev to tacex_fem_set_contact every step (the previous
 * positions survive that call; they are reset when contact is disabled, enabled for the first time, or when tacex_fem_step runs
 * with another workspace / num_envs).  A ROTATION of a capsule or mesh indenter (kinds 3 / 4: the last three row entries) between
 * two steps is not part of the displacement - a spinning indenter drags the pad as if it only translated.
 * Acts inside tacex_fem_step only (tacex_fem_newton_step has no notion of the step's start).  friction_ratio 0 = off. */
int tacex_fem_set_friction(tacex_fem_ctx* ctx, double friction_ratio, double eps_velocity);

/* WHERE the friction lag (normal force lam and normal n per contact vertex, frozen for the time step) is taken:
 *   1 - IPC's rule to the letter (Li et al. 2020, section 5.4, "lagged from the previous time step"): barrier force and normal of the
 *       PREVIOUS configuration, i.e. the positions the step starts from against the indenter where it stood at the previous step;
 *   0 - (round 4) at the step's start positions against the indenter's NEW position, the force capped by the contact reaction there.
 * Both are the previous step's normal force where that step converged tightly and the indenter approaches; they differ when it
 * retreats (0 takes the smaller, already relaxed barrier force) and at loose Newton tolerances.  Only mode 1 makes the step's end
 * state a stationary point of the plain incremental potential of IPC (tests/test_fem_physics_gpu.py); mode 0 (the default) is the
 * one that stays bounded at the reference's default Newton tolerance, where the previous configuration is not in balance. */
int tacex_fem_set_friction_lag(tacex_fem_ctx* ctx, int mode);

/* Which Newton kernel the last tacex_fem_step / tacex_fem_newton_step of the context launched: 1 = CU-resident (the env's state on one CU),
 * 0 = the streaming fallback (larger meshes, the deterministic switch on more than 512 vertices), -1 = none yet.  (ABI 11) */
int tacex_fem_newton_resident(const tacex_fem_ctx* ctx);

/* Contact-following start of tacex_fem_step's Newton loop (default on): a surface vertex inside the barrier zone of the indenter's
 * previous position starts the iteration displaced by the indenter's translation since the previous step (its gap is what it was).
 * An initial guess only - the step's minimiser is unchanged - but the one that lets a RETREATING indenter cost 2-3 Newton iterations
 * like a pressing one instead of 4-30 (libuipc starts from the current positions: world.advance(), US:250-252; 0 restores that). */
int tacex_fem_set_contact_following(tacex_fem_ctx* ctx, int enable);

/* Summation order of the CU-resident Newton kernel's tet -> vertex sums.  0 (default): per-tet rows are added into per-vertex LDS
 * accumulators with ds_add_f64 - the order of a vertex's ~24 contributions depends on wave timing, so two runs agree to round-off
 * (1e-16 relative per add), not bit for bit.  1: rows travel through an exchange window and are gathered in a fixed order (bit-identical
 * runs, ~25 % more time per PCG iteration).  No counterpart in the reference (libuipc's CUDA backend uses atomics throughout). */
int tacex_fem_set_deterministic(tacex_fem_ctx* ctx, int enable);

/* Two-level preconditioner of the Newton system (CU-resident kernel): z = D^-1 r (3x3 block Jacobi, always) + P A_c^-1 P^T r.
 * P: every vertex has 8 (coarse node, weight) pairs - the trilinear hat functions of a small grid laid over the mesh
 * (vertex_nodes_host (V,8) int32 in [0, num_coarse), vertex_weights_host (V,8) f64 >= 0, rows summing to 1; unused slots weight 0);
 * coarse_inverse_host (3 num_coarse, 3 num_coarse) f64 = inverse of P^T A_0 P with A_0 = M (1 + s C) + dt^2 K(rest state), built
 * by the caller (UipcSim does it from tacex_fem_element_terms at the rest state and the constraint flags; constant per mesh
 * and constraint set).  num_coarse <= 64; 0 switches the coarse correction off.  Tables are copied. */
int tacex_fem_set_coarse_space(tacex_fem_ctx* ctx, int num_coarse, const int32_t* vertex_nodes_host,
                               const double* vertex_weights_host, const double* coarse_inverse_host);

/* gaps_dev (num_envs, V) f64 <- signed distance of every vertex of x_dev (num_envs, V, 3) to its env's indenter, by the solver's own
 * distance function (+inf without an indenter): what the caller of tacex_fem_step needs to keep the "an indenter approaches by less
 * than the current gap" contract (UipcSim.contact_gaps). */
int tacex_fem_contact_gaps(tacex_fem_ctx* ctx, const double* x_dev, double* gaps_dev, int num_envs, void* stream);

/* Rigid TRIANGLE-MESH indenter shared by all envs (indenter kind 4 of tacex_fem_set_contact's rows): vertices (num_verts,3) f64 in
 * the mesh's own frame, triangles (num_tris,3) int32.  An env's indenter row [4, px, py, pz, offset, rx, ry, rz] places it: p =
 * position of the mesh origin, r = rotation vector (axis * angle), offset >= 0 inflates the surface.  The barrier acts between
 * every weighted gelpad vertex and its nearest triangle (point-triangle distance, closest feature = face / edge / vertex);
 * the distance is unsigned.  num_tris = 0 removes the mesh.  Tables are copied. */
int tacex_fem_set_indenter_mesh(tacex_fem_ctx* ctx, int num_verts, const double* verts_host, int num_tris, const int32_t* tris_host);

/* Block part of the preconditioner: block-tridiagonal LDL^T along VERTEX CHAINS instead of one 3x3 block per vertex.  A chain is a
 * sequence of mesh vertices, consecutive ones sharing a tet (the columns of vertices through a gelpad's thickness: the nearly
 * incompressible material couples the layers of a thin pad most strongly; UipcSim builds them with
 * coarse_space.build_vertex_chains).  chain_offsets_host (num_chains + 1) indexes chain_vertices_host; every vertex may appear in at
 * most one chain, the rest are chains of one vertex (= block Jacobi).  num_chains = 0 switches chains off.  Per Newton iteration the
 * kernel factors S_0 = D_0, G_i = S_i^-1 A(i, i+1), S_{i+1} = D_{i+1} - A(i, i+1)^T G_i and keeps S^-1 / G as floats in LDS.
 * Tables are copied.  (CU-resident Newton kernel only.) */
int tacex_fem_set_chains(tacex_fem_ctx* ctx, int num_chains, const int32_t* chain_offsets_host, const int32_t* chain_vertices_host);

/* Device-side convergence for repeated Newton launches: dx_dev (num_envs,) f64 holds, per env, max |d| of the UNSCALED Newton
 * direction of its last iteration - not of the update the CCD bound and the line search made of it (the caller fills it with
 * +inf at the start of a time step); an env whose value is <= dx_tol
 * (= velocity_tol * dt, US:62-66) returns at once from the next tacex_fem_newton_step, so extra iterations cost nothing and
 * no host round trip is needed to stop them.  nullptr disables. */
int tacex_fem_set_newton_early_exit(tacex_fem_ctx* ctx, double* dx_dev, double dx_tol);

/* One projected-Newton iteration per env: assemble (block-Jacobi + coarse-grid preconditioned, tacex_fem_set_coarse_space) system, matrix-free PCG
 * (US:70-72 tol_rate: the PCG stops when r^T M^-1 r <= tol_rate x b^T M^-1 b - libuipc's own test, LinearPCG::pcg `abs(rz_new) <=
 * global_tol_rate * rz0`: relative on r.z itself, i.e. sqrt(tol_rate) on the M^-1 norm), backtracking line search on the energy (US:76: max_iter 8; when the capped search finds no decrease
 * the step is halved further, down to 2^-32, instead of leaving the env stuck).  x_dev is updated in place.
 * stats_dev (B,4) float64 = [energy_before, energy_after, step_length, pcg_iterations]. */
int tacex_fem_newton_step(tacex_fem_ctx* ctx, double* x_dev, const double* x_tilde_dev,
                          const uint8_t* constrained_dev, const double* aim_dev, double* stats_dev,
                          void* workspace_dev, int num_envs, int pcg_max_iter, double pcg_tol_rate,
                          int ls_max_iter, void* stream);

/* One backward-Euler time step of every env - what `world.advance()` does for the gelpad (US:250-252) - with NO host round trip:
 *   x_prev = x;  x_tilde = x + dt v + dt^2 gravity;  up to max_newton Newton iterations (as tacex_fem_newton_step), each env leaving
 *   the loop on the device once the UNSCALED Newton direction of an iteration has max |d| <= velocity_tol * dt (US:62-66; IPC's
 *   test on the search direction, whatever the CCD bound and the line search made of the step);
 *   v = (x - x_prev) / dt.
 * x_dev, v_dev (B,V,3) f64 are updated in place, x_tilde_dev (B,V,3) is written.  With the CU-resident Newton kernel (one thread per
 * vertex: 512 threads per env for meshes of <= 512 vertices, 768 threads for larger ones as long as the env's state fits the CU's
 * 160 KB of LDS - about 600 vertices with friction, 745 without; simple_axle.msh, 593 vertices / 2 003 tets, does with friction; the
 * wide variant takes analytic indenters only and is not available with tacex_fem_set_deterministic) the whole loop is ONE launch; the streaming fallback
 * (any vertex count; analytic indenters with barrier, step bound and - ABI 11 - friction lagged at the step's start; block-Jacobi PCG) launches
 * max_newton kernels on a fixed schedule in which converged envs return at once.  stats_dev (B,4) = [energy_before, energy_after, step_length, pcg_iterations] of the LAST iteration
 * run; step_info_dev (B,4) f64 = [newton_iterations, max |d| of the last iteration, flags, pcg_iterations_total] (both kernels: the fallback
 * sums / ORs over its launches), flags: 1 = a contact vertex was at or beyond its indenter's surface when an iteration started
 * (the caller moved the indenter by more than the gap: that vertex gets no restoring force), 2 = a line search found no decrease;
 * informational: 4 = the env dropped the coarse correction for the rest of the step (its stopping test passed with the residual's
 * 2-norm above |b|: no reduction at all), 8 = its PCG met negative curvature and iterations of the step were solved with the PSD-safe Hessian
 * (|c_J| of the Stable Neo-Hookean d2J/dF2 term clamped per element; the gradient is exact, the minimiser the same).
 * workspace_dev: tacex_fem_workspace_bytes(ctx, num_envs). */
int tacex_fem_step(tacex_fem_ctx* ctx, double* x_dev, double* v_dev, double* x_tilde_dev, const uint8_t* constrained_dev,
                   const double* aim_dev, double* stats_dev, double* step_info_dev, void* workspace_dev, int num_envs,
                   const double gravity[3], int max_newton, double velocity_tol, int pcg_max_iter, double pcg_tol_rate,
                   int ls_max_iter, void* stream);

/* ---- The reference's own UIPC scene: a FREE affine-body ball on a ground plane under the gelpad (second slice) --------
 * Replaces, for `UipcObjectCfg(constitution_cfg=AffineBodyConstitutionCfg())` next to the gelpad
 * (scripts/benchmarking/tactile_sim_performance/envs/ball_rolling_uipc.py:71-92; source/tacex_uipc/tacex_uipc/objects/uipc_object.py:62-74,
 * 456-466: `AffineBodyConstitution().apply_to(mesh, m_kappa * MPa, mass_density)`, kinematic = False) and `ground(ground_height,
 * ground_normal)` + the default contact model (source/tacex_uipc/tacex_uipc/sim/uipc_sim.py:192-201), what libuipc's
 * `world.advance()` does with them.  libuipc is not in the reference tree: the model is oracle/abd_oracle.py's (PARITY UNPINNED) -
 * Lan et al. 2022 (affine body: q = (p, A), mass matrix S (x) I_3 from the surface mesh's moments, orthogonality energy
 * kappa vol |A^T A - I|^2), Li et al. 2020 (barrier on every point-triangle pair closer than d_hat, pad vertex / ball triangle AND
 * ball vertex / pad triangle, ground against the surface vertices of both bodies), additive CCD on the pairs, and lagged Coulomb
 * friction of EVERY contact (pairs of both kinds and ground contacts of both bodies; ratio and stick tolerance from tacex_fem_set_friction,
 * US:103-124: the contacts of the state the step starts from are frozen - normal force, normal, barycentric weights - and slide
 * relative to it, Li et al. 2020 section 5.4).  EDGE-EDGE pairs (every pad surface edge against every ball edge closer than d_hat:
 * segment-segment distance, weight = the mean of the two edges' areas, IPC's mollifier m(|e_a x e_b|^2) with threshold
 * 1e-3 |e_a|^2 |e_b|^2 at rest, Li et al. 2020 eq. 24) complete IPC's contact set; tacex_fem_set_edge_edge(ctx, 0) switches that pair kind
 * off (on after tacex_fem_set_affine_body).
 *
 * tacex_fem_set_affine_body: ONE body per env, the same mesh for all envs.  verts_host (num_verts,3) f64 in the body frame, tris_host
 * (num_tris,3) outward oriented; density [kg/m^3]; kappa [Pa] (m_kappa * 1e6); pad_vertex_area_host (V) contact weights of the gelpad's
 * vertices (0: interior) and pad_tris_host (num_pad_tris,3) its surface triangles; d_hat [m], stiffness [J/m^2] as tacex_fem_set_contact;
 * the ground is the half-space z >= ground_height (enable_ground = 0: none); kinematic = 1 (`AffineBodyConstitutionCfg.kinematic`,
 * uipc_object.py:70-73, 463-466 `is_fixed`): the body's twelve unknowns are FIXED within a step - the caller moves q_dev between steps
 * and the pad feels the body through the pairs and their friction.  num_verts = 0 removes the body.  Tables are copied.
 * State: q_dev / qv_dev (num_envs,4,3) f64 = (p, c_1, c_2, c_3) with c_k = COLUMN k of A (a surface point is p + sum_k X_k c_k) and its
 * velocity.  tacex_fem_ball_step = one backward-Euler step of pad + ball: predictor (gravity on the pad vertices and on p), the whole
 * Newton loop in one launch (PCG preconditioned by 3x3 blocks on the pad and the exact 12x12 ball block; an env leaves the loop once
 * the unscaled direction has max |d| <= velocity_tol * dt on the position rows AND <= transrate_tol * dt on the affine rows,
 * uipc_sim.py:62-66), velocities.  step_info (num_envs,4) = [Newton iterations, max |d|, flags (1 a surface vertex at / below the
 * ground, 2 line search failed, 16 a candidate list overflowed), PCG iterations].  workspace: tacex_fem_ball_workspace_bytes.
 * tacex_fem_ball_terms: energy (num_envs) and gradient (num_envs, V + 4, 3) of the step's incremental potential at (x, q) against the
 * predictors (x_tilde, q_tilde) - the entry point the parity tests compare with the oracle term by term.  x_prev_dev / q_prev_dev
 * (both or neither): the state friction slides relative to; the friction lag (forces, normals, weights) is then taken at (x, q) itself.
 * tacex_fem_ball_moments: the 4x4 moment matrix S (row-major) and kappa * vol the library derived from the mesh. */
int tacex_fem_set_affine_body(tacex_fem_ctx* ctx, int num_verts, const double* verts_host, int num_tris, const int32_t* tris_host, double density,
                              double kappa, const double* pad_vertex_area_host, int num_pad_tris, const int32_t* pad_tris_host, double d_hat,
                              double stiffness, double ground_height, int enable_ground, int kinematic);
int tacex_fem_set_edge_edge(tacex_fem_ctx* ctx, int enable);
/* Line search of tacex_fem_ball_step: after a backtracking (halving) search that had to cut the step, `bisections` more energy evaluations
 * between the accepted and the last rejected step keep the LARGEST step that still does not increase the incremental potential.  A cut step
 * was cut by a pair entering the barrier zone (contact resistance 10 GPa against a 0.1 MPa gel: micrometres inside cost more than the step
 * gains); the accepted half usually leaves that pair just outside d_hat, where it has no curvature for the next iteration either, which is cut
 * again - the larger step takes it inside, into the next Hessian.  Same acceptance rule (E <= E0 on a CCD-feasible step), same minimiser;
 * default 4 (bench scene: worst env of a step 4.4 -> 3.1 Newton iterations, 135 K -> 178 K frames/s per 512-env shard), 0 = plain halving. */
int tacex_fem_set_line_search_refine(tacex_fem_ctx* ctx, int bisections);
size_t tacex_fem_ball_workspace_bytes(const tacex_fem_ctx* ctx, int num_envs);
int tacex_fem_ball_moments(const tacex_fem_ctx* ctx, double moments_out[16], double* kappa_vol_out);
int tacex_fem_ball_terms(tacex_fem_ctx* ctx, const double* x_dev, const double* x_tilde_dev, const double* q_dev, const double* q_tilde_dev,
                         const uint8_t* constrained_dev, const double* aim_dev, const double* x_prev_dev, const double* q_prev_dev,
                         double* energy_dev, double* grad_dev, double* step_info_dev, void* workspace_dev, int num_envs, void* stream);
int tacex_fem_ball_step(tacex_fem_ctx* ctx, double* x_dev, double* v_dev, double* q_dev, double* qv_dev, const uint8_t* constrained_dev,
                        const double* aim_dev, double* step_info_dev, void* workspace_dev, int num_envs, const double gravity[3], int max_newton,
                        double velocity_tol, double transrate_tol, int pcg_max_iter, double pcg_tol_rate, int ls_max_iter, void* stream);

/* Per-env reset of the FEM state - what `UipcObject.reset(env_ids)` / `write_vertex_positions_to_sim(vertex_positions, env_ids)`
 * (source/tacex_uipc/tacex_uipc/objects/uipc_object.py:280-370; `reset` is a TODO stub there, the write ignores env_ids) are for: an
 * RL task puts ONE env's gelpad back while the others go on.  For every env of env_ids_dev (num_reset,) int32 (NULL: all num_envs):
 *   x <- positions_dev (num_reset, V, 3) f64 when given, else the rest positions of tacex_fem_create;   v <- 0;
 *   step_info row <- 0 (nullable);   the indenter position the friction displacement of the NEXT tacex_fem_step is measured from
 *   <- none (the env's next step sees no sliding, like the first step of a fresh scene, wherever its indenter is put meanwhile).
 * Nothing else of the solver survives a time step (friction lag, warm start, lagged preconditioner blocks and x_prev are rebuilt
 * by every tacex_fem_step), so the next step of a reset env is bit-identical to the first step of a fresh context in deterministic
 * mode.  workspace_dev: the workspace tacex_fem_step runs with (nullable before the first step).  Enqueued on `stream`. */
int tacex_fem_reset_envs(tacex_fem_ctx* ctx, const int32_t* env_ids_dev, int num_reset, const double* positions_dev, double* x_dev,
                         double* v_dev, double* step_info_dev, void* workspace_dev, int num_envs, void* stream);

/* Attachment animation (UA:364-428: `_compute_aim_positions` + the animator callback `animate_tet` UA:365-385) for all envs:
 *   aim_position[b, idx[a]] = R(body_quat[b]) * offsets[a] + body_pos[b];  is_constrained[b, idx[a]] = 1
 * body_pos_dev (B,3) / body_quat_dev (B,4 wxyz) float32 pose of the rigid body the gelpad is attached to (IsaacLab root / link
 * state), offsets_dev (A,3) float32 attachment offsets in the body frame (UA:293-297), idx_dev (A,) int32 vertex ids.
 * The rotation is evaluated in float32 like IsaacLab's transform_points (UA:411-413) and widened to float64 on store.
 * aim_compact_dev (B,A,3) float64 optionally receives the same positions densely (the reference's `self.aim_positions`). */
int tacex_fem_set_attachment_targets(const float* body_pos_dev, const float* body_quat_dev, const float* offsets_dev,
                                     const int32_t* idx_dev, double* aim_position_dev, uint8_t* is_constrained_dev,
                                     double* aim_compact_dev, int num_envs, int num_attachment_points, int num_verts,
                                     void* stream);

/* FEM-driven markers (VT:347-366): barycentric surface points -> pinhole projection.
 *   surf_pos_dev (B,Vs,3) surface vertex positions in the CAMERA frame, tri_dev (M,3) int32 vertex ids,
 *   weight_dev (M,3) float64, intrinsics fx,fy,cx,cy (VT:57-63: 340,325,160,125) -> uv_dev (B,M,2) float64 */
int tacex_fem_marker_uv(const double* surf_pos_dev, const int32_t* tri_dev, const double* weight_dev,
                        double fx, double fy, double cx, double cy, double* uv_dev, int num_envs,
                        int num_surf_verts, int num_markers, void* stream);

/* gen_marker_flow's per-step part for a static marker grid (VT:354-413: no random rotation / translation / shift / noise / lost
 * tracking - the shipped cfgs) in ONE launch: the FEM state's surface vertices -> camera frame (VT:142-187) -> barycentric point ->
 * pinhole projection of all markers, then the step's subset next to its initial projection.
 *   x_dev (B,V,3) f64 world positions (the FEM state itself); surf_ids_dev (Vs) int64: global vertex id of every surface vertex;
 *   cam_pos_dev (B,3), cam_rot_inv_dev (B,3,3) row-major: camera-frame point = cam_rot_inv (x - cam_pos); tri_dev (M,3) int32 SURFACE-local
 *   vertex ids and weight_dev (M,3) f64 per marker; init_uv_dev (B,M,2) the projections of the reference surface; select_dev (K) int64: the
 *   markers of this step's subset (VT:394-399, drawn by the caller); normalize_div > 0: values / normalize_div - 1 (VT:407-409), 0: pixels.
 *   Outputs: curr_uv_dev (B,M,2) f64 (nullable) all current projections; flow_dev (B,2,K,2) f64 and / or flow_f32_dev (B,2,K,2) f32:
 *   [initial | current] (u, v) of the subset. */
int tacex_fem_marker_flow(const double* x_dev, const int64_t* surf_ids_dev, const double* cam_pos_dev, const double* cam_rot_inv_dev,
                          const int32_t* tri_dev, const double* weight_dev, double fx, double fy, double cx, double cy,
                          const double* init_uv_dev, const int64_t* select_dev, double normalize_div, double* curr_uv_dev, double* flow_dev,
                          float* flow_f32_dev, int num_envs, int num_verts, int num_markers, int num_selected, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TACEX_HIP_H */
