// FOTS marker-displacement field for MI355X (gfx950).
//
// Reference semantics: source/tacex/tacex/simulation_approaches/fots/sim/marker_motion.py (MM) driven per
// env by fots/fots_marker_sim.py:130-182 (FS).  The reference loops over envs in Python with host syncs;
// here all envs are processed by two launches:
//   fots_reduce_kernel : one workgroup per env - max of the deformed gel, contact-mask centroid sums
//                        (wave shuffles + LDS), float4 / uchar4 coalesced loads;
//   fots_marker_kernel : one workgroup (128 threads) per env - global max over envs, contact list by
//                        wave ballot in the reference's (col-major) order, dilate / shear / twist in
//                        float64 like NumPy does (MM:78-120), trajectory state update (FS:168,176-177).
// dtypes follow the reference: depth map / centroid / traj entries are float32, marker grids int,
// displacements float64, output float32.
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <stdint.h>

#include "tacex_hip.h"
#include "tacex_internal.h"

namespace tacex {

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void fots_reduce_kernel(const float* __restrict__ z,
                                                          const uint8_t* __restrict__ mask,
                                                          FotsReduce* __restrict__ out, int H, int W) {
  const int e = blockIdx.x;
  const int npix = H * W;
  const float* zz = z + (size_t)e * npix;
  const uint8_t* mm = mask + (size_t)e * npix;
  float zmax = -INFINITY;
  int cnt = 0, sr = 0, sc = 0;
  const int n4 = npix >> 2;  // W % 4 == 0 is checked on the host, so a group of 4 never straddles rows
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    v4f v = reinterpret_cast<const v4f*>(zz)[i];
    uchar4 m = reinterpret_cast<const uchar4*>(mm)[i];
    zmax = fmaxf(zmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    const int p = i << 2;
    const int row = p / W, col = p - row * W;
    const int c = (m.x != 0) + (m.y != 0) + (m.z != 0) + (m.w != 0);
    cnt += c;
    sr += c * row;
    sc += (m.x != 0) * col + (m.y != 0) * (col + 1) + (m.z != 0) * (col + 2) + (m.w != 0) * (col + 3);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
    cnt += __shfl_xor(cnt, o, 64);
    sr += __shfl_xor(sr, o, 64);
    sc += __shfl_xor(sc, o, 64);
  }
  __shared__ float s_z[16];
  __shared__ int s_c[16], s_r[16], s_k[16];
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_z[wid] = zmax; s_c[wid] = cnt; s_r[wid] = sr; s_k[wid] = sc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) { zmax = fmaxf(zmax, s_z[w]); cnt += s_c[w]; sr += s_r[w]; sc += s_k[w]; }
    out[e].zmax = zmax; out[e].count = cnt; out[e].sum_row = sr; out[e].sum_col = sc;
  }
}

// per-env statistics from the per-wave partials the fused Taxim tail kernel wrote while the frame was still in LDS
// (n partials per env, env-major): replaces the 98 MB re-read of fots_reduce_kernel by a 5 KB one
__global__ __launch_bounds__(64) void fots_combine_kernel(const FotsReduce* __restrict__ part, int n, FotsReduce* __restrict__ out) {
  const int e = blockIdx.x;
  const FotsReduce* p = part + (size_t)e * n;
  float zmax = -INFINITY;
  int cnt = 0, sr = 0, sc = 0;
  for (int i = threadIdx.x; i < n; i += 64) {
    const FotsReduce r = p[i];
    zmax = fmaxf(zmax, r.zmax); cnt += r.count; sr += r.sum_row; sc += r.sum_col;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
    cnt += __shfl_xor(cnt, o, 64);
    sr += __shfl_xor(sr, o, 64);
    sc += __shfl_xor(sc, o, 64);
  }
  if (threadIdx.x == 0) { out[e].zmax = zmax; out[e].count = cnt; out[e].sum_row = sr; out[e].sum_col = sc; }
}

struct FotsArgs {
  const float* z; const uint8_t* mask; const float* indent; const float* theta;
  float* traj; float* markers; const FotsReduce* red;
  const int* mx; const int* my;
  int B, H, W, nrow, ncol;
  int compact;  // z / mask are (B, M) values at the marker pixels instead of (B, H, W) frames
  double lamb0, lamb1, lamb2;
  float mm2pix, shear_max, theta_max_rad_f;
};

__global__ __launch_bounds__(128) void fots_marker_kernel(FotsArgs a) {
  const int e = blockIdx.x;
  const int M = a.nrow * a.ncol;
  const int tid = threadIdx.x;
  __shared__ float s_gmax;
  __shared__ int s_ncontact;
  __shared__ int s_cy[128], s_cx[128];
  __shared__ double s_ch[128];
  __shared__ int s_flag[128];

  // global max over the whole batch (FS:130 `deformed_gel.max()`), re-derived per workgroup from the
  // per-env maxima (B floats, L2 resident)
  float g = -INFINITY;
  for (int i = tid; i < a.B; i += blockDim.x) g = fmaxf(g, a.red[i].zmax);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o, 64));
  __shared__ float s_g2[2];
  if ((tid & 63) == 0) s_g2[tid >> 6] = g;
  __syncthreads();
  if (tid == 0) s_gmax = fmaxf(s_g2[0], s_g2[1]);
  __syncthreads();
  const float gmax = s_gmax;
  const FotsReduce red = a.red[e];
  const size_t fo = (size_t)e * a.H * a.W;
  float* out_init = a.markers + (size_t)e * 2 * M * 2;
  float* out_cur = out_init + (size_t)M * 2;
  float* tr = a.traj + (size_t)e * 8;
  const bool in_contact = a.indent[e] > 0.0f;  // FS:133

  // marker m is stored row-major (row j, col i); the reference visits contacts col-major: for i: for j
  const bool active = tid < M;
  int j = 0, i = 0, px = 0, py = 0;
  if (active) { j = tid / a.ncol; i = tid - j * a.ncol; px = a.mx[tid]; py = a.my[tid]; }
  if (active) { out_init[tid * 2 + 0] = (float)px; out_init[tid * 2 + 1] = (float)py; }

  if (!in_contact) {  // FS:176-179: traj reset, markers = initial positions
    if (active) { out_cur[tid * 2 + 0] = (float)px; out_cur[tid * 2 + 1] = (float)py; }
    if (tid == 0) { tr[0] = 0.0f; tr[7] = 0.0f; }
    return;
  }

  // ---- contact list (MM:152-166) ----
  bool hit = false;
  float hval = 0.0f;
  if (active && py >= 0 && py < a.H && px >= 0 && px < a.W) {
    const size_t pi = a.compact ? (size_t)e * M + tid : fo + (size_t)py * a.W + px;
    if (a.mask[pi] == 1) {
      hit = true;
      // deformed' = gmax - Z (FS:130, fp32); depth = deformed' - min_env(deformed') (MM:146); /= 10 (MM:149)
      const float dz = gmax - a.z[pi];
      const float dmin = gmax - red.zmax;
      hval = (dz - dmin) / 10.0f;
    }
  }
  s_flag[tid] = hit ? 1 : 0;
  __syncthreads();
  // rank of this marker in the reference's visiting order (col-major index = i * nrow + j)
  if (hit) {
    const int key = i * a.nrow + j;
    int rank = 0;
    for (int m2 = 0; m2 < M; ++m2) {
      if (s_flag[m2]) {
        const int j2 = m2 / a.ncol, i2 = m2 - j2 * a.ncol;
        rank += (i2 * a.nrow + j2) < key;
      }
    }
    s_cy[rank] = py; s_cx[rank] = px; s_ch[rank] = (double)hval;
  }
  if (tid == 0) {
    int n = 0;
    for (int m2 = 0; m2 < M; ++m2) n += s_flag[m2];
    s_ncontact = n;
  }
  __syncthreads();
  const int nc = s_ncontact;

  // ---- trajectory state (FS:135-141,168): centroid of the contact mask in mm, float32 like the reference ----
  float t0x, t0y, t0t, tlx, tly, tlt;
  int tlen;
  {
    const float mean_r = (float)((double)red.sum_row / (double)red.count);  // torch.mean(points.float())
    const float mean_c = (float)((double)red.sum_col / (double)red.count);
    const float cy = (mean_r - (float)(a.H / 2.0)) / a.mm2pix;  // FS:139
    const float cx = (mean_c - (float)(a.W / 2.0)) / a.mm2pix;  // FS:141
    const float th = a.theta[e];
    tlen = (int)tr[0];
    if (tlen == 0) { t0x = cx; t0y = cy; t0t = th; } else { t0x = tr[1]; t0y = tr[2]; t0t = tr[3]; }
    tlx = cx; tly = cy; tlt = th;
    tlen += 1;
  }
  __syncthreads();  // everyone has read the old state
  if (tid == 0) {
    tr[0] = (float)tlen; tr[1] = t0x; tr[2] = t0y; tr[3] = t0t; tr[4] = tlx; tr[5] = tly; tr[6] = tlt;
    tr[7] = (float)nc;  // contacts found this step (MM:152-166), for parity checks
  }
  if (!active) return;

  if (nc == 0) {  // MM:168-170: no marker in contact -> initial positions
    out_cur[tid * 2 + 0] = (float)px; out_cur[tid * 2 + 1] = (float)py;
    return;
  }

  // ---- dilate (MM:111-120), float64 ----
  double dx = 0.0, dy = 0.0;
  for (int c = 0; c < nc; ++c) {
    const double ox = (double)(px - s_cx[c]), oy = (double)(py - s_cy[c]);
    const double gg = exp(-a.lamb0 * (ox * ox + oy * oy));
    dx += s_ch[c] * ox * gg;
    dy += s_ch[c] * oy * gg;
  }
  double nx = (double)px + dx, ny = (double)py + dy;

  if (tlen >= 2) {
    // ---- shear (MM:176-187,78-88): float32 products (traj entries are np.float32), int() truncation ----
    const int scx = (int)(t0x * a.mm2pix + (float)(a.W / 2.0));
    const int scy = (int)(t0y * a.mm2pix + (float)(a.H / 2.0));
    int shx = (int)((tlx - t0x) * a.mm2pix);
    int shy = (int)((tly - t0y) * a.mm2pix);
    const int smax = (int)a.shear_max;
    shx = shx < -smax ? -smax : (shx > smax ? smax : shx);
    shy = shy < -smax ? -smax : (shy > smax ? smax : shy);
    {
      const double ox = (double)(px - scx), oy = (double)(py - scy);
      const double gg = exp(-a.lamb1 * (ox * ox + oy * oy));
      nx += (double)shx * gg;
      ny += (double)shy * gg;
    }
    // ---- twist (MM:193-205,90-109): theta and its cos/sin are float32 (np.float32 scalar), cos(theta-1) sic ----
    float theta = tlt - t0t;
    theta = fminf(fmaxf(theta, -a.theta_max_rad_f), a.theta_max_rad_f);
    const int tcx = (int)(tlx * a.mm2pix + (float)(a.W / 2.0));
    const int tcy = (int)(tly * a.mm2pix + (float)(a.H / 2.0));
    const double c1 = (double)cosf(theta - 1.0f);
    const double s1 = (double)sinf(theta);
    const double ox = (double)(px - tcx), oy = (double)(py - tcy);
    const double gg = exp(-a.lamb2 * (ox * ox + oy * oy));
    nx += (ox * c1 - oy * s1) * gg;
    ny += (ox * s1 + oy * c1) * gg;
  }
  out_cur[tid * 2 + 0] = (float)nx;
  out_cur[tid * 2 + 1] = (float)ny;
}


// ------------------------------------------------------------------------------------------------
// Marker image (FS:346-384 `draw_markers`) + RGB x marker overlay (FS:265-272), SURVEY 8(f) n3.
// The reference stamps one 12 x 12 patch per marker into a (H+24, W+24) uint8 canvas of 255s, the patch chosen from a
// pre-drawn table by the sub-pixel phase of the marker centre, LATER MARKERS OVERWRITING EARLIER ONES where stamps
// overlap, and crops the 12-pixel margin.  Here one workgroup owns a band of canvas rows of one env in LDS (the whole
// 264 x 344 canvas of a 320 x 240 image is 90 KB), stamps the markers in index order with a barrier in between (the order
// is the semantics; a stamp is 144 byte writes), then writes the cropped band - and, on request, the overlay
// uint8(float64(rgb32 * 255) * float64(marker) / 255) exactly as NumPy promotes it.
// Pure integer / byte work apart from the two floor()s of the float64 marker coordinates: bit-exact.
// ------------------------------------------------------------------------------------------------
struct MarkerImgArgs {
  const float* markers;   // (B, 2, M, 2) f32: [initial | current] x (x, y); the current positions are drawn
  const uint8_t* patches; // (SR, SR, S, 12, 12) u8
  const float* rgb;       // (B, H, W, 3) f32 in [0,1], nullable
  uint8_t* img;           // (B, H, W) u8, nullable when only the overlay is wanted
  uint8_t* overlay;       // (B, H, W, 3) u8, nullable
  int B, M, H, W, SR, S, patch_w;  // patch_w = floor((marker_size - base_radius) * SR)
  int band_rows, nbands;
};
constexpr int kMkPad = 12, kMkPatch = 12;

__global__ __launch_bounds__(256) void fots_marker_image_kernel(MarkerImgArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t canvas[];  // band_rows x (W + 24)
  const int e = blockIdx.x / a.nbands, band = blockIdx.x - e * a.nbands;
  const int CW = a.W + 2 * kMkPad, CH = a.H + 2 * kMkPad;
  const int y0 = band * a.band_rows, y1 = min(CH, y0 + a.band_rows);  // canvas rows of this band
  const int tid = threadIdx.x;
  for (int i = tid * 4; i < (y1 - y0) * CW; i += 256 * 4) *reinterpret_cast<uint32_t*>(canvas + i) = 0xffffffffu;  // CW % 4 == 0
  __syncthreads();
  const float* cur = a.markers + ((size_t)e * 2 + 1) * a.M * 2;
  for (int m = 0; m < a.M; ++m) {
    // marker_uv + 0.5 (float64, FS:361) + 12 (FS:366-367)
    const double u = ((double)cur[m * 2 + 0] + 0.5) + 12.0, v = ((double)cur[m * 2 + 1] + 0.5) + 12.0;
    const double fu = floor(u), fv = floor(v);
    const int pu = (int)floor((u - fu) * a.SR), pv = (int)floor((v - fv) * a.SR);  // FS:368-369
    const int cu = (int)fu - 6, cv = (int)fv - 6;                                   // FS:375-376
    // FS:377: canvas_w - 12 > cu >= 0 and canvas_h - 12 > cv >= 0 (a NaN / huge coordinate fails the test like in Python)
    const bool ok = u == u && v == v && fabs(u) < 1e9 && fabs(v) < 1e9 && cu >= 0 && cu < CW - 12 && cv >= 0 && cv < CH - 12;
    if (ok && cv + kMkPatch > y0 && cv < y1 && tid < kMkPatch * kMkPatch) {
      const int dy = tid / kMkPatch, dx = tid - dy * kMkPatch, yy = cv + dy;
      if (yy >= y0 && yy < y1)
        canvas[(yy - y0) * CW + cu + dx] =
            a.patches[((((size_t)pu * a.SR + pv) * a.S + a.patch_w) * kMkPatch + dy) * kMkPatch + dx];
    }
    __syncthreads();  // stamps are ordered: marker m + 1 may overwrite marker m
  }
  // cropped rows of this band: canvas rows [max(y0, 12), min(y1, H + 12))
  const int r0 = max(y0, kMkPad), r1 = min(y1, a.H + kMkPad);
  for (int i = tid; i < (r1 - r0) * a.W; i += 256) {
    const int ry = i / a.W, x = i - ry * a.W;
    const uint8_t mv = canvas[(r0 + ry - y0) * CW + kMkPad + x];
    const size_t p = ((size_t)e * a.H + (r0 + ry - kMkPad)) * a.W + x;
    if (a.img) a.img[p] = mv;
    if (a.overlay) {
      // FS:268-271: tactile_rgb (f32) * 255 -> f32; * (marker.astype(f64) / 255) -> f64; astype(uint8) truncates
      const double f = (double)mv / 255.0;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) a.overlay[p * 3 + ch] = (uint8_t)(int)((double)(a.rgb[p * 3 + ch] * 255.0f) * f);
    }
  }
}

}  // namespace tacex

using namespace tacex;

struct tacex_fots_ctx {
  int device = 0;
  int H = 0, W = 0, nrow = 0, ncol = 0;
  int* mx_dev = nullptr;
  int* my_dev = nullptr;
  double lamb[3];
  float mm2pix, shear_max, theta_max_deg;
};

extern "C" {

int tacex_fots_create(int device_id, const tacex_fots_params* p, tacex_fots_ctx** out) {
  if (!p || !out || !p->marker_x || !p->marker_y) { set_error("tacex_fots_create: null argument"); return 2; }
  const int M = p->num_markers_row * p->num_markers_col;
  if (M <= 0 || M > 128) { set_error("tacex_fots_create: %d markers unsupported (1..128)", M); return 2; }
  if (p->width % 4 != 0) { set_error("tacex_fots_create: image width %d must be a multiple of 4", p->width); return 2; }
  hipError_t e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
  auto* c = new tacex_fots_ctx();
  c->device = device_id; c->H = p->height; c->W = p->width; c->nrow = p->num_markers_row; c->ncol = p->num_markers_col;
  for (int k = 0; k < 3; ++k) c->lamb[k] = p->lamb[k];
  c->mm2pix = p->mm2pix; c->shear_max = p->shear_max; c->theta_max_deg = p->theta_max_deg;
  if ((e = hipMalloc((void**)&c->mx_dev, M * sizeof(int))) != hipSuccess ||
      (e = hipMalloc((void**)&c->my_dev, M * sizeof(int))) != hipSuccess ||
      (e = hipMemcpy(c->mx_dev, p->marker_x, M * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(c->my_dev, p->marker_y, M * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) {
    tacex_fots_destroy(c);
    return fail_hip(e, "tacex_fots_create: table upload");
  }
  *out = c;
  return 0;
}

void tacex_fots_destroy(tacex_fots_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->mx_dev) (void)hipFree(c->mx_dev);
  if (c->my_dev) (void)hipFree(c->my_dev);
  delete c;
}

size_t tacex_fots_state_bytes(int num_envs) { return num_envs > 0 ? (size_t)num_envs * 8 * sizeof(float) : 0; }
size_t tacex_fots_workspace_bytes(int num_envs) { return num_envs > 0 ? (size_t)num_envs * sizeof(FotsReduce) : 0; }

static int fots_markers_impl(tacex_fots_ctx* c, const float* z, const uint8_t* mask, const float* indent, const float* theta,
                             float* traj_state, float* markers, void* ws, const FotsReduce* part, int npart, int B, void* stream, int compact = 0);

int tacex_fots_markers(tacex_fots_ctx* c, const float* z, const uint8_t* mask, const float* indent,
                       const float* theta, float* traj_state, float* markers, void* ws, int B, void* stream) {
  if (!c || !z || !mask || !indent || !theta || !traj_state || !markers || !ws) {
    set_error("tacex_fots_markers: null argument");
    return 2;
  }
  if (B <= 0) return 0;
  return fots_markers_impl(c, z, mask, indent, theta, traj_state, markers, ws, nullptr, 0, B, stream);
}

int tacex_fots_markers_partials(tacex_fots_ctx* c, const float* z, const uint8_t* mask, const float* indent,
                                const float* theta, float* traj_state, float* markers, void* ws, const void* partials,
                                int partials_per_env, int B, void* stream) {
  if (!c || !z || !mask || !indent || !theta || !traj_state || !markers || !ws || !partials || partials_per_env < 1) {
    set_error("tacex_fots_markers_partials: null argument");
    return 2;
  }
  if (B <= 0) return 0;
  return fots_markers_impl(c, z, mask, indent, theta, traj_state, markers, ws, static_cast<const FotsReduce*>(partials),
                           partials_per_env, B, stream);
}

int tacex_fots_markers_compact(tacex_fots_ctx* c, const float* z_pix, const uint8_t* mask_pix, const float* indent,
                               const float* theta, float* traj_state, float* markers, void* ws, const void* partials,
                               int partials_per_env, int B, void* stream) {
  if (!c || !z_pix || !mask_pix || !indent || !theta || !traj_state || !markers || !ws || !partials || partials_per_env < 1) {
    set_error("tacex_fots_markers_compact: null argument");
    return 2;
  }
  if (B <= 0) return 0;
  return fots_markers_impl(c, z_pix, mask_pix, indent, theta, traj_state, markers, ws, static_cast<const FotsReduce*>(partials),
                           partials_per_env, B, stream, 1);
}

static int fots_markers_impl(tacex_fots_ctx* c, const float* z, const uint8_t* mask, const float* indent, const float* theta,
                             float* traj_state, float* markers, void* ws, const FotsReduce* part, int npart, int B, void* stream, int compact) {
  hipStream_t st = (hipStream_t)stream;
  FotsReduce* red = static_cast<FotsReduce*>(ws);
  if (part) hipLaunchKernelGGL(fots_combine_kernel, dim3(B), dim3(64), 0, st, part, npart, red);
  else hipLaunchKernelGGL(fots_reduce_kernel, dim3(B), dim3(1024), 0, st, z, mask, red, c->H, c->W);
  FotsArgs a{};
  a.compact = compact;
  a.z = z; a.mask = mask; a.indent = indent; a.theta = theta; a.traj = traj_state; a.markers = markers; a.red = red;
  a.mx = c->mx_dev; a.my = c->my_dev; a.B = B; a.H = c->H; a.W = c->W; a.nrow = c->nrow; a.ncol = c->ncol;
  a.lamb0 = c->lamb[0]; a.lamb1 = c->lamb[1]; a.lamb2 = c->lamb[2];
  a.mm2pix = c->mm2pix; a.shear_max = c->shear_max;
  a.theta_max_rad_f = (float)(c->theta_max_deg / 180.0 * 3.14159265358979323846);
  hipLaunchKernelGGL(fots_marker_kernel, dim3(B), dim3(128), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "fots kernels");
  return 0;
}

int tacex_fots_marker_image(const float* markers, const uint8_t* patches, int super_res, int size_slots, int patch_w,
                            const float* rgb, uint8_t* img, uint8_t* overlay, int B, int M, int H, int W, void* stream) {
  if (!markers || !patches || (!img && !overlay)) { set_error("tacex_fots_marker_image: null argument"); return 2; }
  if (overlay && !rgb) { set_error("tacex_fots_marker_image: the overlay needs the RGB frame"); return 2; }
  if (super_res < 1 || size_slots < 1 || patch_w < 0 || patch_w >= size_slots) {
    set_error("tacex_fots_marker_image: patch table index %d outside [0, %d)", patch_w, size_slots);
    return 2;
  }
  if (W % 4 != 0 || H <= 0 || W <= 0 || M < 0) { set_error("tacex_fots_marker_image: need W %% 4 == 0, H > 0"); return 2; }
  if (B <= 0) return 0;
  MarkerImgArgs a{};
  a.markers = markers; a.patches = patches; a.rgb = rgb; a.img = img; a.overlay = overlay;
  a.B = B; a.M = M; a.H = H; a.W = W; a.SR = super_res; a.S = size_slots; a.patch_w = patch_w;
  const int CW = W + 2 * kMkPad, CH = H + 2 * kMkPad;
  const size_t budget = 96 * 1024;  // canvas bytes per workgroup
  a.band_rows = (int)std::min<size_t>((size_t)CH, budget / CW);
  if (a.band_rows < 2 * kMkPatch) { set_error("tacex_fots_marker_image: image too wide (%d)", W); return 2; }
  a.nbands = (CH + a.band_rows - 1) / a.band_rows;
  const size_t lds = (size_t)a.band_rows * CW;
  auto kern = fots_marker_image_kernel;
  if (lds > 64 * 1024) {
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
      if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(fots_marker_image_kernel)");
      attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(B * a.nbands), dim3(256), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fots_marker_image_kernel");
}

}  // extern "C"
