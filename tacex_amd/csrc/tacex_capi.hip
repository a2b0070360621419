// C ABI of libtacex_hip.so (include/tacex_hip.h): context management, table upload, stage sequencing.
// No torch types, no allocation inside compute calls (tables are allocated once at *_create).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <mutex>
#include <vector>

#include "tacex_hip.h"
#include "tacex_internal.h"

namespace tacex {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail_hip(hipError_t e, const char* what) {
  set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  return 1;
}

}  // namespace tacex

using namespace tacex;

#define HIP_TRY(expr, what)                          \
  do {                                               \
    hipError_t _e = (expr);                          \
    if (_e != hipSuccess) return fail_hip(_e, what); \
  } while (0)

struct tacex_taxim_ctx {
  int device = 0;
  int H = 0, W = 0;
  int n_levels = 0;
  int n_fused = 0;       // trailing levels handled by the fused tail kernel (+ shading)
  bool use_tail = true;
  bool use_stream = true;  // prefer the streaming tail where it exists (tacex_taxim_set_fused_tail(ctx, 2) forces the tiled one)
  float contact_scale = 0.4f;
  LevelDesc levels[TACEX_MAX_LEVELS];
  ShadeParams shade;
  ShadowParams shadow;
  float* gel_dev = nullptr;
  FotsReduce* fots_part = nullptr;  // optional per-wave FOTS contact statistics written by the fused tail
  int fots_cap = 0;                 // frames the buffer holds
  FotsTaps fots_taps{};             // marker pixels per tile (tacex_taxim_set_fots_taps)
  float* fots_pix_z = nullptr;      // (cap, M) deformed gel at the marker pixels
  uint8_t* fots_pix_m = nullptr;    // (cap, M) contact mask at the marker pixels
  int fots_pix_cap = 0;
  ObsTables obs_tab{};   // filter tables of the last policy-observation size asked for (built on first use)
  // host copies of the row / column filters (the streaming tail's per-row tables are derived from them)
  std::vector<int> obs_ylo_h, obs_ycnt_h, obs_xlo_h, obs_xcnt_h;
  std::vector<float> obs_wy_h;
  // streaming tail: geometry + per-(observation size, segment count) tables, marker pixels as a CSR over rows
  struct StreamObsPlan { int oh, ow, nseg, mk_version; StreamPlan plan; };
  std::vector<StreamObsPlan> stream_plans;
  std::vector<float> feat_y_h;                 // host copy of the polynomial feature y (per-row table of the streaming tail)
  std::vector<int> mk_row_ptr_h;               // marker CSR over frame rows (host), device columns / ids below
  const int* mk_x = nullptr; const int* mk_id = nullptr;
  std::vector<int> mk_x_h, mk_id_h;
  int mk_version = 0;
  const int* frame_rows = nullptr; int frame_rows_cap = 0;  // caller-owned contact row ranges (tacex_taxim_set_frame_rows)
  int* stream_order = nullptr; int stream_order_cap = 0;  // item order of a streaming-tail launch (stream_order_kernel), grown by stream_plan
  // depth -> height map pass handed to the NEXT render of this context (tacex_taxim_defer_height_map_from_depth): the render runs it
  // per band-level chunk on the chunk's own stream, so chunk k+1's depth pass (HBM-bound) overlaps chunk k's band levels (matrix pipe)
  struct DepthPass {
    bool armed = false;
    const float* depth = nullptr; float near_mm = 0, far_m = 0, far_mm = 0, gelpad_h = 0, gelpad_dmin = 0;
    float* hm = nullptr; float* fmin = nullptr; float* indent = nullptr; uint8_t* cam_u8 = nullptr; int* rows = nullptr;
    int B = 0;
  } depth_pass;
  // second stream of the band levels (pipeline_impl: odd chunks of a pass run beside the even ones), created on first use
  static constexpr int kMaxLvlStreams = 4;
  hipStream_t lvl_stream[kMaxLvlStreams - 1] = {}; hipStream_t lvl_caller = nullptr; hipEvent_t lvl_fork = nullptr, lvl_join[kMaxLvlStreams - 1] = {}, order_evt = nullptr;
  std::vector<void*> allocs;
  // profiling
  bool profiling = false;
  struct Ev { hipEvent_t a, b; int stage; };
  std::vector<Ev> events;
  std::vector<double> stage_ms;
  std::vector<int> stage_n;
  std::vector<std::string> stage_names;
};

template <typename T>
static int upload(tacex_taxim_ctx* c, const T* host, size_t n, T** dev) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, n * sizeof(T)), "hipMalloc(table)");
  c->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(table)");
  *dev = static_cast<T*>(p);
  return 0;
}

extern "C" {

const char* tacex_last_error(void) { return g_err; }
int tacex_abi_version(void) { return TACEX_ABI_VERSION; }

int tacex_device_count(int* count) {
  HIP_TRY(hipGetDeviceCount(count), "hipGetDeviceCount");
  return 0;
}

int tacex_device_arch(int device_id, char* buf, size_t buflen) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id), "hipGetDeviceProperties");
  snprintf(buf, buflen, "%s", prop.gcnArchName);
  return 0;
}

int tacex_taxim_create(int device_id, const tacex_taxim_params* p, tacex_taxim_ctx** out) {
  if (!p || !out) { set_error("tacex_taxim_create: null argument"); return 2; }
  if (p->n_levels < 1 || p->n_levels > TACEX_MAX_LEVELS) { set_error("n_levels %d out of range", p->n_levels); return 2; }
  if (p->height < 3 || p->width < 3) { set_error("image %dx%d too small", p->height, p->width); return 2; }
  if (p->num_bins < 2 || p->num_bins > 256) { set_error("num_bins %d out of range (2..256)", p->num_bins); return 2; }
  for (int l = 0; l < p->n_levels; ++l) {
    if (p->ksize_w[l] % 2 != 1 || p->ksize_h[l] % 2 != 1 || p->ksize_w[l] < 1 || p->ksize_h[l] < 1) {
      set_error("level %d: kernel sizes must be odd and positive (%d, %d)", l, p->ksize_w[l], p->ksize_h[l]);
      return 2;
    }
    // torch 'reflect' padding requires pad < dim (TT:411)
    if ((p->ksize_w[l] - 1) / 2 >= p->width || (p->ksize_h[l] - 1) / 2 >= p->height) {
      set_error("level %d: reflect padding (%d, %d) must be smaller than the image (%d, %d)", l,
                (p->ksize_w[l] - 1) / 2, (p->ksize_h[l] - 1) / 2, p->width, p->height);
      return 2;
    }
  }
  HIP_TRY(hipSetDevice(device_id), "hipSetDevice");
  auto* c = new tacex_taxim_ctx();
  c->device = device_id;
  c->H = p->height;
  c->W = p->width;
  c->n_levels = p->n_levels;
  c->contact_scale = p->contact_scale;
  const size_t npix = (size_t)c->H * c->W;
  int rc = 0;
  for (int l = 0; l < p->n_levels && !rc; ++l) {
    c->levels[l].kw = p->ksize_w[l];
    c->levels[l].kh = p->ksize_h[l];
    c->levels[l].same_taps = p->ksize_w[l] == p->ksize_h[l] &&
                             memcmp(p->taps_w[l], p->taps_h[l], sizeof(float) * p->ksize_w[l]) == 0;
    rc |= upload(c, p->taps_w[l], (size_t)p->ksize_w[l], &c->levels[l].taps_w_dev);
    rc |= upload(c, p->taps_h[l], (size_t)p->ksize_h[l], &c->levels[l].taps_h_dev);
    if (c->levels[l].same_taps && !rc) {
      // per-lane band weights of blur_mfma_kernel (v_mfma_f32_16x16x4_f32), lane group g = lane >> 4, i = lane & 15:
      // table 0 (V-pass): k-step ks contracts window index u = 4 ks + g; table 1 (H-pass): u = mfma_h_window_col(KS, g, ks)
      // (four consecutive k-steps = four consecutive window columns = one ds_read_b128, the chunks dealt out to the lane
      // groups so that the read is free of LDS bank conflicts); weight = w[u - RA - i + R], 0 outside the band
      const int K = p->ksize_w[l], R = (K - 1) / 2, RA = (R + 7) & ~7, ks = (16 + 2 * RA) / 4;
      std::vector<float> tl((size_t)2 * ks * 64, 0.0f);
      for (int tab = 0; tab < 2; ++tab)
        for (int kk = 0; kk < ks; ++kk)
          for (int ln = 0; ln < 64; ++ln) {
            const int u = tab == 0 ? 4 * kk + (ln >> 4) : mfma_h_window_col(ks, ln >> 4, kk);
            const int t = u - RA - (ln & 15) + R;
            if (t >= 0 && t < K) tl[((size_t)tab * ks + kk) * 64 + ln] = p->taps_w[l][t];
          }
      rc |= upload(c, tl.data(), tl.size(), &c->levels[l].taps_mfma_dev);
    }
    if (c->levels[l].same_taps && !rc) {
      std::vector<float> pad((size_t)p->ksize_w[l] + 64, 0.0f);
      for (int k = 0; k < p->ksize_w[l]; ++k) pad[16 + k] = p->taps_w[l][k];
      rc |= upload(c, pad.data(), pad.size(), &c->levels[l].taps_pad_dev);
    }
  }
  if (!rc) rc |= upload(c, p->gel_map, npix, &c->gel_dev);
  {
    bool gz = true;
    for (size_t i = 0; i < npix && gz; ++i) gz = p->gel_map[i] == 0.0f;
    for (int l = 0; l < p->n_levels; ++l) c->levels[l].gel_zero = gz;
  }
  // polynomial table (3, nb, nb, 6) -> (nb, nb, 24): one 96-byte, 16-byte-aligned record per bin
  const int nb = p->num_bins;
  if (!rc) {
    std::vector<float> poly((size_t)nb * nb * 24, 0.0f);
    for (int ch = 0; ch < 3; ++ch)
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j)
          for (int k = 0; k < 6; ++k)
            poly[((size_t)i * nb + j) * 24 + (ch == 2 ? 12 : ch * 6) + k] =
                p->poly[(((size_t)ch * nb + i) * nb + j) * 6 + k];
    rc |= upload(c, poly.data(), poly.size(), &c->shade.poly_dev);
  }
  if (!rc) {  // background (3,H,W) -> (H,W,3)
    std::vector<float> bg(npix * 3);
    for (int ch = 0; ch < 3; ++ch)
      for (size_t i = 0; i < npix; ++i) bg[i * 3 + ch] = p->background[ch * npix + i];
    rc |= upload(c, bg.data(), bg.size(), &c->shade.bg_nhwc_dev);
  }
  if (!rc) rc |= upload(c, p->feat_x, (size_t)c->W, &c->shade.fx_dev);
  if (!rc) rc |= upload(c, p->feat_y, (size_t)c->H, &c->shade.fy_dev);
  c->feat_y_h.assign(p->feat_y, p->feat_y + c->H);
  if (rc) { tacex_taxim_destroy(c); return rc; }
  c->shade.H = c->H; c->shade.W = c->W; c->shade.nb = nb;
  c->shade.calib_h = p->calib_height; c->shade.calib_w = p->calib_width;
  c->shade.pixmm = p->pixmm;
  // bin widths as float32 of the python doubles (TT:243-244)
  c->shade.x_binr = (float)(0.5 * 3.14159265358979323846 / (nb - 1));
  c->shade.y_binr = (float)(2.0 * 3.14159265358979323846 / (nb - 1));
  c->n_fused = tail_levels(c->levels, c->n_levels, c->H, c->W);
  {  // RGB of the undeformed gel (flat rows / flat waves of the streaming tail copy it instead of evaluating the polynomial)
    void* pf = nullptr;
    hipError_t ef = hipMalloc(&pf, (size_t)c->H * c->W * 3 * sizeof(float));
    if (ef != hipSuccess) { tacex_taxim_destroy(c); return fail_hip(ef, "hipMalloc(flat image)"); }
    c->allocs.push_back(pf);
    c->shade.flat_rgb_dev = static_cast<float*>(pf);
    ef = run_stream_flat_image(&c->shade, c->H, c->W, nullptr);
    if (ef == hipSuccess) ef = hipDeviceSynchronize();
    if (ef != hipSuccess) { tacex_taxim_destroy(c); return fail_hip(ef, "stream_flat_image_kernel"); }
  }
  const int ns = p->n_levels + 3;
  c->stage_ms.assign(ns, 0.0);
  c->stage_n.assign(ns, 0);
  c->stage_names.resize(ns);
  c->stage_names[0] = "frame_min";
  for (int l = 0; l < p->n_levels; ++l) {
    char nm[64];
    snprintf(nm, sizeof(nm), "blur_l%d_k%dx%d", l, p->ksize_w[l], p->ksize_h[l]);
    c->stage_names[1 + l] = nm;
  }
  c->stage_names[ns - 2] = "shade";
  c->stage_names[ns - 1] = "tail_fused";
  *out = c;
  return 0;
}

void tacex_taxim_destroy(tacex_taxim_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& q : c->lvl_stream)
    if (q) (void)hipStreamSynchronize(q);  // (the streams belong to the per-device pool: level_stream())
  if (c->lvl_fork) (void)hipEventDestroy(c->lvl_fork);
  if (c->order_evt) (void)hipEventDestroy(c->order_evt);
  for (auto& e : c->lvl_join)
    if (e) (void)hipEventDestroy(e);
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Triangle-filter taps of torchvision's antialiased bilinear resize (GelSightSensor's 32 x 32 policy observation),
// one axis: the same fp32 expressions as resize_v_kernel / the oracle's resize_bilinear_aa.
static void obs_axis_table(int n_in, int n_out, int k, std::vector<int>& lo, std::vector<int>& cnt, std::vector<float>& sum,
                           std::vector<float>& w) {
  const float sc = (float)n_in / (float)n_out;
  const float sup = sc >= 1.0f ? sc : 1.0f, inv = sc >= 1.0f ? 1.0f / sc : 1.0f;
  lo.resize(n_out); cnt.resize(n_out); sum.resize(n_out); w.assign((size_t)n_out * k, 0.0f);
  for (int o = 0; o < n_out; ++o) {
    const float center = sc * ((float)o + 0.5f);
    const int a = std::max(0, (int)(center - sup + 0.5f)), b = std::min(n_in, (int)(center + sup + 0.5f));
    lo[o] = a; cnt[o] = b - a;
    float s = 0.0f;
    for (int y = a; y < b; ++y) {
      const float t = ((float)y - center + 0.5f) * inv;
      const float v = fmaxf(0.0f, 1.0f - fabsf(t));
      w[(size_t)o * k + (y - a)] = v;
      s += v;
    }
    sum[o] = s;
  }
}

static int ensure_obs_tables(tacex_taxim_ctx* c, int oh, int ow) {
  if (c->obs_tab.oh == oh && c->obs_tab.ow == ow) return 0;
  HIP_TRY(hipSetDevice(c->device), "hipSetDevice");
  ObsTables t{};
  t.oh = oh; t.ow = ow;
  t.ky = 2 * (int)ceilf((float)c->H / oh) + 2;
  t.kx = 2 * (int)ceilf((float)c->W / ow) + 2;
  std::vector<int> lo, cnt; std::vector<float> sum, w;
  int *dlo, *dcnt; float *dsum, *dw;
  obs_axis_table(c->H, oh, t.ky, lo, cnt, sum, w);
  c->obs_ylo_h = lo; c->obs_ycnt_h = cnt; c->obs_wy_h = w;
  if (int rc = upload(c, lo.data(), lo.size(), &dlo)) return rc;
  if (int rc = upload(c, cnt.data(), cnt.size(), &dcnt)) return rc;
  if (int rc = upload(c, sum.data(), sum.size(), &dsum)) return rc;
  if (int rc = upload(c, w.data(), w.size(), &dw)) return rc;
  t.ylo = dlo; t.ycnt = dcnt; t.ysum = dsum; t.wy = dw;
  t.ymax = *std::max_element(cnt.begin(), cnt.end());
  obs_axis_table(c->W, ow, t.kx, lo, cnt, sum, w);
  c->obs_xlo_h = lo; c->obs_xcnt_h = cnt;
  if (int rc = upload(c, lo.data(), lo.size(), &dlo)) return rc;
  if (int rc = upload(c, cnt.data(), cnt.size(), &dcnt)) return rc;
  if (int rc = upload(c, sum.data(), sum.size(), &dsum)) return rc;
  if (int rc = upload(c, w.data(), w.size(), &dw)) return rc;
  t.xlo = dlo; t.xcnt = dcnt; t.xsum = dsum; t.wx = dw;
  t.xmax = *std::max_element(cnt.begin(), cnt.end());
  c->obs_tab = t;
  return 0;
}

// Streaming-tail plan for B frames (+ an optional policy observation of oh x ow): strip / segment geometry, the packed
// per-frame-row table the kernel walks (feature y, which <= 3 observation rows the row feeds with which weights, its FOTS
// marker range) and the block geometry of the observation partial sums.  Cached per (oh, ow, nseg, marker set);
// obs_ready = false when the filters do not fit (a frame row feeding more than 3 observation rows, i.e. down-sampling
// factors below 2, or column windows that exceed the per-wave LDS table).
static int stream_plan(tacex_taxim_ctx* c, int n_fused, int B, int oh, int ow, const StreamPlan** out) {
  const int k0 = c->levels[c->n_levels - n_fused].kw;
  int nstrips = 0, strip_w = 0, lv_nstrips = 0, lv_strip_w = 0;
  if (!stream_geometry(n_fused, k0, c->W, &nstrips, &strip_w, &lv_nstrips, &lv_strip_w)) { set_error("no streaming tail for this level set"); return 1; }
  // two resident rounds of waves per launch (2 or 3 waves per SIMD each, depending on the kernel's register budget)
  const int per_simd = 2 * stream_waves_per_simd(n_fused, k0);
  const int nseg = stream_segments(B, nstrips, c->H, stream_warm_rows(n_fused, k0, false), per_simd);
  const int lv_nseg = stream_segments(B, lv_nstrips, c->H, stream_warm_rows(n_fused, k0, true), per_simd);
  if (B * nstrips * nseg > c->stream_order_cap) {  // item-order buffer of the launch (first call with a larger batch only)
    HIP_TRY(hipSetDevice(c->device), "hipSetDevice");
    void* pbuf = nullptr;
    HIP_TRY(hipMalloc(&pbuf, (size_t)B * nstrips * nseg * sizeof(int)), "hipMalloc(stream order)");
    c->allocs.push_back(pbuf);  // the smaller buffer stays until the context goes (an earlier launch may still read it)
    c->stream_order = static_cast<int*>(pbuf);
    c->stream_order_cap = B * nstrips * nseg;
  }
  for (auto& e : c->stream_plans)
    if (e.oh == oh && e.ow == ow && e.nseg == nseg && e.plan.lv_nseg == lv_nseg && e.mk_version == c->mk_version) { *out = &e.plan; return 0; }
  HIP_TRY(hipSetDevice(c->device), "hipSetDevice");
  StreamPlan p{};
  p.nstrips = nstrips; p.strip_w = strip_w; p.nseg = nseg; p.seg_rows = (c->H + nseg - 1) / nseg;
  p.lv_nstrips = lv_nstrips; p.lv_strip_w = lv_strip_w; p.lv_nseg = lv_nseg; p.lv_seg_rows = (c->H + lv_nseg - 1) / lv_nseg;
  if (lv_nstrips * lv_nseg > (int)(tail_tiles_per_frame(c->H, c->W) * kTailWavesPerTile)) { set_error("streaming tail: FOTS partial slots exceeded"); return 1; }
  const int H = c->H, W = c->W;
  std::vector<StreamRowInfo> rows(H);
  for (int r = 0; r < H; ++r) {
    StreamRowInfo& ri = rows[r];
    ri.fy = c->feat_y_h[r]; ri.o0 = 0; ri.w0 = ri.w1 = ri.w2 = 0.0f; ri.pad = 0;
    ri.mk0 = c->mk_row_ptr_h.empty() ? 0 : c->mk_row_ptr_h[r];
    ri.mk1 = c->mk_row_ptr_h.empty() ? 0 : c->mk_row_ptr_h[r + 1];
  }
  if (oh > 0 && ow > 0) {
    if (int rc = ensure_obs_tables(c, oh, ow)) return rc;
    const int ky = c->obs_tab.ky;
    std::vector<int> oa(nseg), ob(nseg), q0(nstrips), nq(nstrips);
    bool ok = true;
    for (int r = 0; r < H && ok; ++r) {
      int o = 0;
      while (o < oh && r >= c->obs_ylo_h[o] + c->obs_ycnt_h[o]) ++o;
      rows[r].o0 = o;
      float wk[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < 3 && o + k < oh; ++k)
        if (r >= c->obs_ylo_h[o + k] && r < c->obs_ylo_h[o + k] + c->obs_ycnt_h[o + k])
          wk[k] = c->obs_wy_h[(size_t)(o + k) * ky + (r - c->obs_ylo_h[o + k])];
      rows[r].w0 = wk[0]; rows[r].w1 = wk[1]; rows[r].w2 = wk[2];
      if (o + 3 < oh && r >= c->obs_ylo_h[o + 3]) ok = false;  // a 4th observation row overlaps this frame row
      if (r > 0 && rows[r].o0 < rows[r - 1].o0) ok = false;
    }
    int nrows = 0, ncols = 0;
    for (int g = 0; g < nseg && ok; ++g) {
      const int r0 = g * p.seg_rows, r1 = std::min(H, r0 + p.seg_rows);
      oa[g] = rows[r0].o0;
      int b = oa[g];
      while (b + 1 < oh && c->obs_ylo_h[b + 1] < r1) ++b;
      ob[g] = std::min(b, oh - 1);
      if (oa[g] >= oh) ok = false;
      nrows = std::max(nrows, ob[g] - oa[g] + 1);
    }
    for (int s2 = 0; s2 < nstrips && ok; ++s2) {
      const int vx0 = s2 * strip_w, vx1 = std::min(W, vx0 + strip_w);
      int a = 0;
      while (a < ow && c->obs_xlo_h[a] + c->obs_xcnt_h[a] <= vx0) ++a;
      int b = a;
      while (b + 1 < ow && c->obs_xlo_h[b + 1] < vx1) ++b;
      if (a >= ow) ok = false;
      q0[s2] = a; nq[s2] = b - a + 1;
      ncols = std::max(ncols, nq[s2]);
    }
    const int kxp = (c->obs_tab.xmax + 3) & ~3;
    if (ncols > stream_obs_max_cols() || ncols * kxp > stream_obs_lds_floats() || kxp > 64 * 3) ok = false;
    if (ok) {
      int *d_oa, *d_ob, *d_q0, *d_nq;
      if (int rc = upload(c, oa.data(), oa.size(), &d_oa)) return rc;
      if (int rc = upload(c, ob.data(), ob.size(), &d_ob)) return rc;
      if (int rc = upload(c, q0.data(), q0.size(), &d_q0)) return rc;
      if (int rc = upload(c, nq.data(), nq.size(), &d_nq)) return rc;
      p.obs = c->obs_tab; p.obs_seg_oa = d_oa; p.obs_seg_ob = d_ob; p.obs_kxp = kxp;
      p.obs_strip_q0 = d_q0; p.obs_strip_nq = d_nq; p.obs_nrows = nrows; p.obs_ncols = ncols; p.obs_ready = true;
    } else {
      for (auto& ri : rows) { ri.o0 = 0; ri.w0 = ri.w1 = ri.w2 = 0.0f; }
    }
  }
  // device table: [row][record (8 ints) | packed marker slots]
  std::vector<int> table((size_t)H * kStreamRowInts, -1);
  p.mk_vec = !c->mk_row_ptr_h.empty();
  for (int r = 0; r < H; ++r) {
    static_assert(sizeof(StreamRowInfo) == 8 * sizeof(int), "row record layout");
    memcpy(&table[(size_t)r * kStreamRowInts], &rows[r], sizeof(StreamRowInfo));
    if (c->mk_row_ptr_h.empty()) continue;
    const int e0 = c->mk_row_ptr_h[r], e1 = c->mk_row_ptr_h[r + 1];
    if (e1 - e0 > kStreamMkSlots) { p.mk_vec = false; continue; }
    for (int e = e0; e < e1; ++e)
      table[(size_t)r * kStreamRowInts + 8 + (e - e0)] = (int)((unsigned)c->mk_x_h[e] | ((unsigned)c->mk_id_h[e] << 16));
  }
  if (c->W >= 65535) p.mk_vec = false;
  int* d_rows = nullptr;
  if (int rc = upload(c, table.data(), table.size(), &d_rows)) return rc;
  p.rows = d_rows;
  p.mk_x = c->mk_x; p.mk_id = c->mk_id; p.n_markers = c->fots_taps.n_markers;
  if (p.n_markers >= 65535) p.mk_vec = false;
  c->stream_plans.push_back({oh, ow, nseg, c->mk_version, p});
  *out = &c->stream_plans.back().plan;
  return 0;
}

int tacex_taxim_set_shadow(tacex_taxim_ctx* c, const tacex_shadow_params* p) {
  if (!c || !p || !p->fan_angles || !p->fan_cos || !p->fan_sin || !p->table || !p->blur_taps_w || !p->blur_taps_h) { set_error("tacex_taxim_set_shadow: null argument"); return 2; }
  if (p->num_directions < 1 || p->num_fan_rays < 1 || p->num_heights < 2 || p->num_steps < 1 || p->blur_kw % 2 != 1 || p->blur_kh % 2 != 1) {
    set_error("tacex_taxim_set_shadow: bad table dimensions");
    return 2;
  }
  HIP_TRY(hipSetDevice(c->device), "hipSetDevice");
  ShadowParams& s = c->shadow;
  s.ndir = p->num_directions; s.nfan = p->num_fan_rays; s.nheight = p->num_heights; s.nstep = p->num_steps;
  s.wl = p->win_left; s.wr = p->win_right; s.wt = p->win_top; s.wb = p->win_bottom;
  s.depth0 = p->shadow_depth_0; s.height_prec = p->height_precision; s.disc_prec = p->discretize_precision;
  s.step_x = p->step_x; s.step_y = p->step_y;
  int rc = upload(c, p->fan_angles, (size_t)s.ndir * s.nfan, &s.fan_dev);
  if (!rc) rc |= upload(c, p->fan_cos, (size_t)s.ndir * s.nfan, &s.fan_cos_dev);
  if (!rc) rc |= upload(c, p->fan_sin, (size_t)s.ndir * s.nfan, &s.fan_sin_dev);
  if (!rc) {  // (3, ndir, nh, nstep) -> (ndir, nh, nstep, 4)
    std::vector<float> t((size_t)s.ndir * s.nheight * s.nstep * 4, 0.0f);
    for (int ch = 0; ch < 3; ++ch)
      for (int d = 0; d < s.ndir; ++d)
        for (int h = 0; h < s.nheight; ++h)
          for (int k = 0; k < s.nstep; ++k)
            t[(((size_t)d * s.nheight + h) * s.nstep + k) * 4 + ch] = p->table[(((size_t)ch * s.ndir + d) * s.nheight + h) * s.nstep + k];
    rc |= upload(c, t.data(), t.size(), &s.table_dev);
  }
  s.sblur_kw = p->blur_kw; s.sblur_kh = p->blur_kh;
  if (!rc) rc |= upload(c, p->blur_taps_w, (size_t)p->blur_kw, &s.sblur_taps_w_dev);
  if (!rc) rc |= upload(c, p->blur_taps_h, (size_t)p->blur_kh, &s.sblur_taps_h_dev);
  const LevelDesc& fl = c->levels[c->n_levels - 1];  // deform_final_sigma kernel (TT:343-344)
  s.final_kw = fl.kw; s.final_kh = fl.kh; s.final_taps_w_dev = fl.taps_w_dev; s.final_taps_h_dev = fl.taps_h_dev;
  if (rc) return rc;
  s.ready = true;
  return 0;
}

size_t tacex_taxim_shadow_workspace_bytes(const tacex_taxim_ctx* c, int B) {
  if (!c || B <= 0) return 0;
  const size_t img = align_up((size_t)B * c->H * c->W * sizeof(float), 256);
  return 12 * img;  // deformed gel 1, mask 1 (u8, one image slot), gdir 1, raw 3, shadow 3, tmp 3
}

int tacex_taxim_shadow_rays(tacex_taxim_ctx* c, const float* z, const uint8_t* mask, const float* gdir, float* shadow_min, int B,
                            void* stream) {
  if (!c || !z || !mask || !gdir || !shadow_min) { set_error("tacex_taxim_shadow_rays: null argument"); return 2; }
  if (!c->shadow.ready) { set_error("tacex_taxim_shadow_rays: needs tacex_taxim_set_shadow first"); return 2; }
  if (B <= 0) return 0;
  HIP_TRY(run_shadow_rays(c->shadow, c->shade, z, mask, c->gel_dev, gdir, shadow_min, B, (hipStream_t)stream), "shadow_ray_kernel");
  return 0;
}

size_t tacex_taxim_workspace_bytes(const tacex_taxim_ctx* c, int B) {
  if (!c || B <= 0) return 0;
  const size_t img = align_up((size_t)B * c->H * c->W * sizeof(float), 256);
  const size_t vec = align_up((size_t)B * sizeof(float), 256);
  return 3 * img + 3 * vec + align_up((size_t)B * 4 * sizeof(int), 256);  // Z ping, Z pong, generic-path temp; shift_a, shift_b, pdepth; contact rows / columns
}
// where the library keeps the contact row ranges it computes itself (behind everything a chunk of <= B frames lays out)
static int* workspace_rows(const tacex_taxim_ctx* c, void* ws, int B) {
  const size_t img = align_up((size_t)B * c->H * c->W * sizeof(float), 256);
  const size_t vec = align_up((size_t)B * sizeof(float), 256);
  return reinterpret_cast<int*>(static_cast<char*>(ws) + 3 * img + 3 * vec);
}

int tacex_taxim_set_profiling(tacex_taxim_ctx* c, int enabled) {
  if (!c) { set_error("null ctx"); return 2; }
  c->profiling = enabled != 0;
  return 0;
}

int tacex_taxim_num_stages(const tacex_taxim_ctx* c) { return c ? (int)c->stage_ms.size() : 0; }

const char* tacex_taxim_stage_name(const tacex_taxim_ctx* c, int stage) {
  if (!c || stage < 0 || stage >= (int)c->stage_names.size()) return "";
  return c->stage_names[stage].c_str();
}

static void drain_events(tacex_taxim_ctx* c) {
  for (auto& e : c->events) {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      c->stage_ms[e.stage] += ms;
      c->stage_n[e.stage] += 1;
    }
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  c->events.clear();
}

int tacex_taxim_read_profile(tacex_taxim_ctx* c, int stage, double* total_ms, int* launches) {
  if (!c || stage < 0 || stage >= (int)c->stage_ms.size()) { set_error("bad stage"); return 2; }
  drain_events(c);
  if (total_ms) *total_ms = c->stage_ms[stage];
  if (launches) *launches = c->stage_n[stage];
  c->stage_ms[stage] = 0.0;
  c->stage_n[stage] = 0;
  return 0;
}

struct StageTimer {
  tacex_taxim_ctx* c; hipStream_t st; int stage; hipEvent_t a{}, b{}; bool on;
  StageTimer(tacex_taxim_ctx* c_, hipStream_t st_, int stage_) : c(c_), st(st_), stage(stage_), on(c_->profiling) {
    if (on) {
      on = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess &&
           hipEventRecord(a, st) == hipSuccess;
    }
  }
  ~StageTimer() {
    if (on && hipEventRecord(b, st) == hipSuccess) c->events.push_back({a, b, stage});
  }
};

}  // extern "C"

// frames [b0, b0 + nb) of a depth -> height map pass (GS:581-593 + TS:115-131 + contact row / column ranges)
static int depth_pass_range(const tacex_taxim_ctx::DepthPass& d, int b0, int nb, int H, int W, hipStream_t st) {
  const size_t o = (size_t)b0 * H * W;
  uint8_t* u8 = d.cam_u8 ? d.cam_u8 + o : nullptr;
  float* ind = d.indent ? d.indent + b0 : nullptr;
  if (d.rows && frame_rows_supported(H, W)) {
    HIP_TRY(run_frame_rows(d.depth + o, true, d.hm + o, d.fmin + b0, ind, u8, nullptr, d.rows + 4 * b0, nb, H, W, d.near_mm, d.far_m, d.far_mm,
                           d.gelpad_h, d.gelpad_dmin, st),
            "frame_rows_kernel<depth>");
    return 0;
  }
  if (d.rows) HIP_TRY(run_fill_rows(d.rows + 4 * b0, nb, H, W, st), "fill_rows_kernel");
  HIP_TRY(run_frame_min(d.depth + o, true, d.hm + o, d.fmin + b0, ind, u8, nb, H * W, d.near_mm, d.far_m, d.far_mm, d.gelpad_h, d.gelpad_dmin, st),
          "frame_min_kernel<depth>");
  return 0;
}

static int fill_depth_pass(tacex_taxim_ctx::DepthPass* d, const char* who, const float* depth_m, double near_m, double far_m, float gelpad_h,
                           float gelpad_dmin, float* hm_mm, float* frame_min, float* indent_mm, uint8_t* cam_u8, int32_t* frame_rows, int B) {
  if (!depth_m || !hm_mm || !frame_min) { set_error("%s: null buffer", who); return 2; }
  if (frame_rows && !indent_mm) { set_error("%s: frame_rows needs indent_mm", who); return 2; }
  d->depth = depth_m;
  // GS:573-574: `clipping_range[i] * 1000` is a Python double product; torch rounds it ONCE to float32 as the scalar operand
  d->near_mm = (float)(near_m * 1000.0); d->far_m = (float)far_m; d->far_mm = (float)(far_m * 1000.0);
  d->gelpad_h = gelpad_h; d->gelpad_dmin = gelpad_dmin;
  d->hm = hm_mm; d->fmin = frame_min; d->indent = indent_mm; d->cam_u8 = cam_u8; d->rows = frame_rows; d->B = B;
  return 0;
}

extern "C" {

int tacex_height_map_from_depth(const float* depth_m, double near_m, double far_m, float gelpad_h,
                                float gelpad_dmin, float* hm_mm, float* frame_min, float* indent_mm,
                                uint8_t* cam_u8, int32_t* frame_rows, int B, int H, int W, void* stream) {
  tacex_taxim_ctx::DepthPass d;
  if (int rc = fill_depth_pass(&d, "tacex_height_map_from_depth", depth_m, near_m, far_m, gelpad_h, gelpad_dmin, hm_mm, frame_min, indent_mm,
                               cam_u8, frame_rows, B)) return rc;
  if (B <= 0) return 0;
  return depth_pass_range(d, 0, B, H, W, (hipStream_t)stream);
}

int tacex_taxim_defer_height_map_from_depth(tacex_taxim_ctx* c, const float* depth_m, double near_m, double far_m, float gelpad_h,
                                            float gelpad_dmin, float* hm_mm, float* frame_min, float* indent_mm, uint8_t* cam_u8,
                                            int32_t* frame_rows, int B) {
  if (!c) { set_error("tacex_taxim_defer_height_map_from_depth: null context"); return 2; }
  if (c->depth_pass.armed) { set_error("tacex_taxim_defer_height_map_from_depth: a deferred pass is already pending on this context"); return 2; }
  if (int rc = fill_depth_pass(&c->depth_pass, "tacex_taxim_defer_height_map_from_depth", depth_m, near_m, far_m, gelpad_h, gelpad_dmin, hm_mm,
                               frame_min, indent_mm, cam_u8, frame_rows, B)) return rc;
  c->depth_pass.armed = B > 0;
  return 0;
}

int tacex_taxim_flush_deferred(tacex_taxim_ctx* c, void* stream) {
  if (!c) { set_error("tacex_taxim_flush_deferred: null context"); return 2; }
  if (!c->depth_pass.armed) return 0;
  c->depth_pass.armed = false;
  return depth_pass_range(c->depth_pass, 0, c->depth_pass.B, c->H, c->W, (hipStream_t)stream);
}

int tacex_height_map_from_indenters(const float* indenters, float pixmm, float gel_top_mm, float far_clip_mm, float gelpad_h,
                                    float gelpad_dmin, float* hm_mm, float* frame_min, float* indent_mm, int B, int H, int W,
                                    void* stream) {
  if (!indenters || !hm_mm || !frame_min) { set_error("tacex_height_map_from_indenters: null buffer"); return 2; }
  if (W % 4 != 0 || H <= 0 || !(pixmm > 0.0f)) { set_error("tacex_height_map_from_indenters: need W %% 4 == 0, H > 0, pixmm > 0"); return 2; }
  if (B <= 0) return 0;
  HIP_TRY(run_indenter_height_map(indenters, hm_mm, frame_min, indent_mm, B, H, W, pixmm, gel_top_mm, far_clip_mm, gelpad_h,
                                  gelpad_dmin, (hipStream_t)stream),
          "indenter_height_map_kernel");
  return 0;
}

int tacex_indentation_depth(const float* hm_mm, float gelpad_h, float gelpad_dmin, float* frame_min,
                            float* indent_mm, int32_t* frame_rows, int B, int H, int W, void* stream) {
  if (!hm_mm || !frame_min || !indent_mm) { set_error("tacex_indentation_depth: null buffer"); return 2; }
  if (B <= 0) return 0;
  if (frame_rows && frame_rows_supported(H, W)) {
    HIP_TRY(run_frame_rows(hm_mm, false, nullptr, frame_min, indent_mm, nullptr, nullptr, frame_rows, B, H, W, 0.f, 0.f, 0.f, gelpad_h,
                           gelpad_dmin, (hipStream_t)stream),
            "frame_rows_kernel");
    return 0;
  }
  if (frame_rows) HIP_TRY(run_fill_rows(frame_rows, B, H, W, (hipStream_t)stream), "fill_rows_kernel");
  HIP_TRY(run_frame_min(hm_mm, false, nullptr, frame_min, indent_mm, nullptr, B, H * W, 0.f, 0.f, 0.f, gelpad_h,
                        gelpad_dmin, (hipStream_t)stream),
          "frame_min_kernel");
  return 0;
}

static int pipeline_chunk(tacex_taxim_ctx* c, const float* hm, const float* press, float* frame_min, float* rgb,
                          float* z_out, uint8_t* mask_out, void* ws, int B, unsigned flags, hipStream_t st,
                          float* obs_h, void* obs, int obs_hh, int obs_w, FotsReduce* fots_part, int frame0,
                          const int* rows = nullptr, const tacex_taxim_ctx::DepthPass* dp = nullptr);

// The extra streams the band levels of a pass alternate on (TWO CHUNKS IN FLIGHT, pipeline_chunk) come from ONE pool per device,
// created at first use and kept for the life of the process.  A stream per CONTEXT (round 4) made what a context measured depend on
// how many streams the process had created before it: HIP deals streams onto a handful of hardware queues round-robin, and a
// context whose level stream lands on the queue of the caller's stream runs its "two chunks in flight" one after the other
// (bench.py sweep, 640x480: 110 K frames/s as the 17th rig of a process against 128 K on its own; profiles/r05_experiments.md section 7).
// Round 6: the pool is keyed by (device, caller stream) - contexts driven on DIFFERENT caller streams (bench --sensor-streams, multi-threaded
// hosts) get different side streams, so one context's fork wait does not serialise the other's chunks; all contexts on one caller stream
// (every test, the bench default) still share lane 0, which is what made measurements independent of context creation order.
static hipError_t level_stream(int device, hipStream_t caller, int q, hipStream_t* out) {
  constexpr int kLanes = 4;
  static std::mutex mu;
  static hipStream_t pool[64][kLanes][tacex_taxim_ctx::kMaxLvlStreams - 1] = {};
  static hipStream_t owner[64][kLanes] = {};
  static int n_owner[64] = {};
  if (device < 0 || device >= 64 || q < 0 || q >= tacex_taxim_ctx::kMaxLvlStreams - 1) return hipErrorInvalidValue;
  std::lock_guard<std::mutex> lock(mu);
  int lane = -1;
  for (int l = 0; l < n_owner[device] && l < kLanes; ++l)
    if (owner[device][l] == caller) { lane = l; break; }
  if (lane < 0) {
    lane = n_owner[device] % kLanes;  // (beyond kLanes distinct caller streams the lanes are shared round-robin)
    if (n_owner[device] < kLanes) owner[device][lane] = caller;
    ++n_owner[device];
  }
  if (!pool[device][lane][q]) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != device) { if (hipError_t e = hipSetDevice(device); e != hipSuccess) return e; }
    hipError_t e = hipStreamCreateWithFlags(&pool[device][lane][q], hipStreamNonBlocking);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return e;
  }
  *out = pool[device][lane][q];
  return hipSuccess;
}

// Frames per pass of the pipeline.  With the LDS-tiled tail, large shards are walked in chunks whose level buffers (Z ping /
// pong, 4 B/px each) plus height map stay resident in the 256 MB Infinity Cache (measured in round 1 at 2048 frames: k=33
// 66 -> 58 us per 256 frames).  The streaming tail wants the opposite: one wave marches down a whole strip, so a pass needs
// >= 2048 strips to fill the chip without splitting strips into short segments (each segment re-runs an 18-row warm-up), and
// with it the band kernels measured the same chunked or not (C3: 3.95 ms chunked at 256, 3.86 ms in one pass) - passes of up
// to 1024 frames of 320x240 there.  TACEX_CHUNK_FRAMES overrides (0 = whole batch in one pass).
static int chunk_frames(const tacex_taxim_ctx* c, int B) {
  static const int env = getenv("TACEX_CHUNK_FRAMES") ? atoi(getenv("TACEX_CHUNK_FRAMES")) : -1;
  if (env == 0) return B;
  if (env > 0) return env < B ? env : B;
  const bool stream = c->use_tail && c->use_stream && c->n_fused > 0 &&
                      stream_supported(c->n_fused, c->levels[c->n_levels - c->n_fused].kw, c->H, c->W);
  const size_t per_frame = (size_t)3 * c->H * c->W * sizeof(float);
  // (2048 frames of 320x240 per streaming pass since round 5: a GelSightSensorGroup of two 1024-env sensors is one pass - 1571 us for the
  //  tail's 2048 frames against 2 x 802, C3 660 K against 652 K frames/s in two passes, profiles/r05_experiments.md.
  //  Frames beyond 320x240 keep the 1024-frame-equivalent pass: 640x480 measured 9.28 ms per 1024 frames in passes of 512 against 7.97 ms in passes of 256.)
  const size_t n = stream ? ((size_t)(per_frame <= (size_t)240 * 320 * 12 ? 2048 : 1024) * 240 * 320 * 12) / per_frame : ((size_t)240 << 20) / per_frame;
  if (n < 1 || (size_t)B < (stream ? n + n / 2 : 2 * n)) return B;  // a shard barely over the budget is cheaper in one pass than in two small ones
  const size_t nchunks = ((size_t)B + n - 1) / n;
  return (int)(((size_t)B + nchunks - 1) / nchunks);  // equal chunks
}

// Frames per launch of the band levels inside one pass that ends in the streaming tail (TACEX_LEVEL_CHUNK_FRAMES overrides,
// 0 = the whole pass): 3 buffers x 4 B/px per frame against the 256 MB Infinity Cache.
static int level_chunk_frames(const tacex_taxim_ctx* c, int B) {
  static const int env = getenv("TACEX_LEVEL_CHUNK_FRAMES") ? atoi(getenv("TACEX_LEVEL_CHUNK_FRAMES")) : -1;
  if (env == 0) return B;
  if (env > 0) return env < B ? env : B;
  const size_t per_frame = (size_t)3 * c->H * c->W * sizeof(float);
  const size_t n = ((size_t)240 << 20) / per_frame;
  if (n < 1 || (size_t)B < n + n / 2) return B;
  const size_t nchunks = ((size_t)B + n - 1) / n;
  return (int)(((size_t)B + nchunks - 1) / nchunks);
}

int tacex_taxim_chunk_frames(const tacex_taxim_ctx* c, int B) { return (c && B > 0) ? chunk_frames(c, B) : 0; }

static int pipeline_impl(tacex_taxim_ctx* c, const float* hm, const float* press, float* frame_min, float* rgb,
                         float* z_out, uint8_t* mask_out, void* ws, int B, unsigned flags, hipStream_t st,
                         float* obs_h = nullptr, void* obs = nullptr, int obs_hh = 0, int obs_w = 0) {
  const int cf = chunk_frames(c, B);
  // A deferred depth -> height map pass (tacex_taxim_defer_height_map_from_depth) that produces exactly this call's inputs runs inside
  // the pass, chunk by chunk (pipeline_chunk); one that does not is run in full first - it was promised no later than this call.
  tacex_taxim_ctx::DepthPass dpl;
  const tacex_taxim_ctx::DepthPass* dp = nullptr;
  if (c->depth_pass.armed) {
    c->depth_pass.armed = false;
    dpl = c->depth_pass;
    const bool fits = dpl.B == B && dpl.hm == hm && dpl.fmin == frame_min && dpl.indent == press && press &&
                      (flags & TACEX_FLAG_HAVE_FRAME_MIN) && !(flags & TACEX_FLAG_NO_SHIFT) &&
                      (!(flags & TACEX_FLAG_HAVE_FRAME_ROWS) || dpl.rows == c->frame_rows);
    if (fits) dp = &dpl;
    else if (int rc = depth_pass_range(dpl, 0, dpl.B, c->H, c->W, st)) return rc;
  }
  FotsReduce* fp = (c->fots_part && B <= c->fots_cap) ? c->fots_part : nullptr;
  const size_t fper = tail_tiles_per_frame(c->H, c->W) * kTailWavesPerTile;
  // Contact row ranges (zero-band skipping of the band levels, TACEX_BAND_SKIP=0 disables): from the caller's buffer when the
  // minimum came with them, else from the library's own minimum pass (kept behind the chunk layouts of the workspace)
  static const bool band_skip = !(getenv("TACEX_BAND_SKIP") && atoi(getenv("TACEX_BAND_SKIP")) == 0);
  const bool can_rows = band_skip && press && !(flags & TACEX_FLAG_NO_SHIFT) && frame_rows_supported(c->H, c->W);
  const int* rows = nullptr;
  if (flags & TACEX_FLAG_HAVE_FRAME_MIN) {
    if (can_rows && (flags & TACEX_FLAG_HAVE_FRAME_ROWS) && c->frame_rows && B <= c->frame_rows_cap) rows = c->frame_rows;
  } else {  // one reduction pass over the whole shard
    StageTimer t(c, st, 0);
    if (can_rows) {
      int* wr = workspace_rows(c, ws, B);
      HIP_TRY(run_frame_rows(hm, false, nullptr, frame_min, nullptr, nullptr, press, wr, B, c->H, c->W, 0.f, 0.f, 0.f, 0.f, 0.f, st),
              "frame_rows_kernel");
      rows = wr;
    } else {
      HIP_TRY(run_frame_min(hm, false, nullptr, frame_min, nullptr, nullptr, B, c->H * c->W, 0.f, 0.f, 0.f, 0.f, 0.f, st),
              "frame_min_kernel");
    }
    flags |= TACEX_FLAG_HAVE_FRAME_MIN;
  }
  if (cf >= B)
    return pipeline_chunk(c, hm, press, frame_min, rgb, z_out, mask_out, ws, B, flags, st, obs_h, obs, obs_hh, obs_w, fp,
                          B <= c->fots_pix_cap ? 0 : -1, rows, dp);
  const size_t npix = (size_t)c->H * c->W;
  for (int b0 = 0; b0 < B; b0 += cf) {
    const int n = B - b0 < cf ? B - b0 : cf;
    tacex_taxim_ctx::DepthPass dpc;
    if (dp) {  // the chunk's share of the deferred pass
      dpc = *dp;
      dpc.depth += b0 * npix; dpc.hm += b0 * npix; dpc.fmin += b0; dpc.indent += b0; dpc.B = n;
      if (dpc.cam_u8) dpc.cam_u8 += b0 * npix;
      if (dpc.rows) dpc.rows += 4 * b0;
    }
    int rc = pipeline_chunk(c, hm + b0 * npix, press ? press + b0 : nullptr, frame_min + b0, rgb ? rgb + b0 * npix * 3 : nullptr,
                            z_out ? z_out + b0 * npix : nullptr, mask_out ? mask_out + b0 * npix : nullptr, ws, n,
                            flags, st, obs_h,
                            obs ? static_cast<char*>(obs) + (size_t)b0 * obs_hh * obs_w * 3 * ((flags & TACEX_FLAG_OBS_U8) ? 1 : 4) : nullptr,
                            obs_hh, obs_w, fp ? fp + (size_t)b0 * fper : nullptr, B <= c->fots_pix_cap ? b0 : -1,
                            rows ? rows + 4 * b0 : nullptr, dp ? &dpc : nullptr);
    if (rc) return rc;
  }
  return 0;
}

// two-pass antialiased down-sample of the finished frame (scratch = [resize temp | float observation when the caller wants uint8])
static int resize_obs(tacex_taxim_ctx* c, const float* rgb, float* scratch, void* obs, bool u8, int oh, int ow, int B, hipStream_t st) {
  const size_t tmp_floats = (size_t)B * (size_t)(c->H * ow > oh * c->W ? c->H * ow : oh * c->W) * 3;
  float* dst = u8 ? scratch + tmp_floats : static_cast<float*>(obs);
  HIP_TRY(run_resize_aa(rgb, c->H, c->W, dst, oh, ow, B, 3, scratch, st), "resize_aa (observation)");
  if (u8) HIP_TRY(run_obs_to_u8(dst, static_cast<uint8_t*>(obs), (size_t)B * oh * ow * 3, st), "obs_to_u8_kernel");
  return 0;
}

static int pipeline_chunk(tacex_taxim_ctx* c, const float* hm, const float* press, float* frame_min, float* rgb,
                          float* z_out, uint8_t* mask_out, void* ws, int B, unsigned flags, hipStream_t st,
                          float* obs_h, void* obs, int obs_hh, int obs_w, FotsReduce* fots_part, int frame0,
                          const int* rows, const tacex_taxim_ctx::DepthPass* dp) {
  const bool obs_u8 = (flags & TACEX_FLAG_OBS_U8) != 0;
  const size_t img = align_up((size_t)B * c->H * c->W * sizeof(float), 256);
  const size_t vec = align_up((size_t)B * sizeof(float), 256);
  char* w = static_cast<char*>(ws);
  float* zbuf[2] = {reinterpret_cast<float*>(w), reinterpret_cast<float*>(w + img)};
  float* tmp = reinterpret_cast<float*>(w + 2 * img);
  const float* sa = reinterpret_cast<float*>(w + 3 * img);
  const float* sb = reinterpret_cast<float*>(w + 3 * img + vec);
  const float* pd = reinterpret_cast<float*>(w + 3 * img + 2 * vec);
  const bool no_shift = (flags & TACEX_FLAG_NO_SHIFT) != 0;
  if (!(flags & TACEX_FLAG_HAVE_FRAME_MIN)) {
    StageTimer t(c, st, 0);
    HIP_TRY(run_frame_min(hm, false, nullptr, frame_min, nullptr, nullptr, B, c->H * c->W, 0.f, 0.f, 0.f, 0.f, 0.f, st),
            "frame_min_kernel");
  }
  if (no_shift) {  // S = hm, P = -min(hm): zeros / negated minima need their own (B,) arrays
    HIP_TRY(run_press_depth(frame_min, press, const_cast<float*>(sa), const_cast<float*>(sb), const_cast<float*>(pd), B, 1, st),
            "press_depth_kernel");
  } else {
    // S = (hm - min) - press (TT:441) and P = -min(S) = -((min - min) - press) = press (TT:449): the per-frame scalars
    // the kernels read ARE the frame-min and press arrays - no (B,)-sized helper launch (~5 us of a ~600 us step)
    sa = frame_min; sb = press; pd = press;
  }
  const int n_fused = c->use_tail ? c->n_fused : 0;
  const int n_band = c->n_levels - n_fused;
  const float* src = nullptr;
  const bool stream_tail = n_fused > 0 && c->use_stream && rgb && !z_out && !mask_out &&
                           stream_supported(n_fused, c->levels[c->n_levels - n_fused].kw, c->H, c->W);
  // Band levels ahead of a streaming tail run over sub-ranges of the pass (level_chunk_frames): the tail wants >= 2048 strips
  // per launch, the band kernels want their three frame-sized buffers (height map, Z ping, Z pong) in the Infinity Cache.
  int lcf = stream_tail ? level_chunk_frames(c, B) : B;
  const size_t npix = (size_t)c->H * c->W;
  // TWO CHUNKS IN FLIGHT: the chunks of a pass are independent until the tail, and a band-level launch of 128-256 frames spends
  // ~14 us of its 40-77 us ramping up and draining (k = 61: 45.5 us for 128 frames, 76.9 for 256).  Chunks therefore run round-robin on the
  // caller's stream and on streams of the context's own - at a chunk size divided by the stream count, so that the frames in flight (and
  // their three buffers in the Infinity Cache) stay what they were.  Single-kernel (matrix-core) levels only: the two-pass
  // fallback shares one scratch image.  Not while profiling: the stage timers bracket launches on the caller's stream.
  static const int lvl_streams_env = getenv("TACEX_LEVEL_STREAMS") ? atoi(getenv("TACEX_LEVEL_STREAMS")) : 2;
  const int lvl_streams = lvl_streams_env > tacex_taxim_ctx::kMaxLvlStreams ? tacex_taxim_ctx::kMaxLvlStreams : lvl_streams_env;
  // (a pass of fewer than three chunks gains 1.4 % - 512 frames of 320x240 - and loses it again when the caller overlaps the pass with
  //  other work of its own, the FEM step of C4: there the extra stream only adds contention.  Three and more: 2.7 % at 1024 frames of
  //  320x240, 8.5 % at 640x480.)
  bool dual = stream_tail && lvl_streams > 1 && !c->profiling && n_band > 0 && lcf < B && (B + lcf - 1) / lcf >= 3;
  for (int l = 0; dual && l < n_band; ++l)
    dual = blur_level_single_kernel(c->levels[l], l == 0, c->H, c->W);
  if (dual) {
    if (!c->lvl_fork) HIP_TRY(hipEventCreateWithFlags(&c->lvl_fork, hipEventDisableTiming), "hipEventCreate");
    for (int q = 0; q < lvl_streams - 1; ++q) {
      if (!c->lvl_stream[q] || c->lvl_caller != st)
        if (hipError_t es = level_stream(c->device, st, q, &c->lvl_stream[q]); es != hipSuccess) return fail_hip(es, "hipStreamCreate(band levels)");
      if (!c->lvl_join[q]) HIP_TRY(hipEventCreateWithFlags(&c->lvl_join[q], hipEventDisableTiming), "hipEventCreate");
    }
    c->lvl_caller = st;
    static const int env_lcf = getenv("TACEX_LEVEL_CHUNK_FRAMES") ? atoi(getenv("TACEX_LEVEL_CHUNK_FRAMES")) : -1;
    if (env_lcf < 0) lcf = (lcf + lvl_streams - 1) / lvl_streams;
  }
  // A deferred depth pass is interleaved with the band levels only where that pays: chunks alternating on two streams, and chunks of
  // >= 128 frames (the pass runs one workgroup per frame).  Measured (profiles/r05_experiments.md section 15): 320x240 in 128-frame chunks
  // +3.2 % on C3; 640x480 in its 32-frame chunks -11 %, in units of 128-1024 frames -0.5..-4 %.  Otherwise: in full, ahead of the levels.
  static const int depth_min_frames = getenv("TACEX_DEPTH_INTERLEAVE_MIN_FRAMES") ? atoi(getenv("TACEX_DEPTH_INTERLEAVE_MIN_FRAMES")) : 128;
  if (dp && (!dual || lcf < depth_min_frames)) {
    if (int rc = depth_pass_range(*dp, 0, B, c->H, c->W, st)) return rc;
    dp = nullptr;
  }
  if (dual) {
    HIP_TRY(hipEventRecord(c->lvl_fork, st), "hipEventRecord");  // the pass's inputs (height map, shifts, rows) are ready behind this
    for (int q = 0; q < lvl_streams - 1; ++q) HIP_TRY(hipStreamWaitEvent(c->lvl_stream[q], c->lvl_fork, 0), "hipStreamWaitEvent");
  }
  hipStream_t const st_main = st;
  // ITEM ORDER OF THE TAIL, EARLY: stream_order_kernel needs the contact rows only.  On the two-streams path it is issued beside the band
  // levels - right behind the fork when the caller brought the rows, behind the last two depth passes when they are this pass's own
  // (deferred) - instead of between the levels' join and the tail (9.5 us + the join's ~12 us of queue turnaround on the critical path,
  // profiles/r05_experiments.md section 22).  TACEX_STREAM_ORDER_EARLY=0: as before.
  static const int order_early_env = getenv("TACEX_STREAM_ORDER_EARLY") ? atoi(getenv("TACEX_STREAM_ORDER_EARLY")) : 1;
  const StreamPlan* tail_plan = nullptr;
  bool order_done = false;
  int band_grow_rows = 0;
  for (int l = 0; l < n_band; ++l) band_grow_rows += (c->levels[l].kh - 1) / 2;
  const int n_chunks = (B + lcf - 1) / lcf;
  bool order_early = order_early_env != 0 && stream_tail && dual && lvl_streams == 2 && rows != nullptr && n_chunks >= 3;
  if (order_early) {
    const bool want_obs_p = obs_h && obs;
    if (int rc = stream_plan(c, n_fused, B, want_obs_p ? obs_hh : 0, want_obs_p ? obs_w : 0, &tail_plan)) return rc;
    if (B * tail_plan->nstrips * tail_plan->nseg > c->stream_order_cap) order_early = false;
  }
  if (order_early && !dp) {  // rows from the caller: ready behind the fork; on the second stream, ahead of its first chunk
    HIP_TRY(run_stream_order(c->levels, c->n_levels, n_fused, *tail_plan, B, c->H, rows, band_grow_rows, c->stream_order, c->lvl_stream[0], &order_done),
            "stream_order_kernel");
  }
  int chunk_no = 0;
  for (int b0 = 0; b0 < B; b0 += lcf, ++chunk_no) {
    const int nb = B - b0 < lcf ? B - b0 : lcf;
    const int lane_q = dual ? chunk_no % lvl_streams : 0;
    hipStream_t st = lane_q > 0 ? c->lvl_stream[lane_q - 1] : st_main;  // (shadows the pass's stream inside the chunk)
    if (dp) {  // this chunk's height maps, minima, indentation depths and contact ranges: on the chunk's stream, ahead of its levels
      if (int rc = depth_pass_range(*dp, b0, nb, c->H, c->W, st)) return rc;
      if (order_early && chunk_no == n_chunks - 2) {  // the last depth pass of THIS stream: the other stream's order launch waits for it
        if (!c->order_evt) HIP_TRY(hipEventCreateWithFlags(&c->order_evt, hipEventDisableTiming), "hipEventCreate");
        HIP_TRY(hipEventRecord(c->order_evt, st), "hipEventRecord");
      } else if (order_early && chunk_no == n_chunks - 1) {  // every frame's rows are written once this one and the other stream's last are done
        HIP_TRY(hipStreamWaitEvent(st, c->order_evt, 0), "hipStreamWaitEvent");
        HIP_TRY(run_stream_order(c->levels, c->n_levels, n_fused, *tail_plan, B, c->H, rows, band_grow_rows, c->stream_order, st, &order_done),
                "stream_order_kernel");
      }
    }
    src = nullptr;
    int grow = 0, grow_x = 0;  // rows / columns by which the non-zero range of the level's input exceeds the contact rows / columns
    for (int l = 0; l < n_band; ++l) {
      const bool last = l == c->n_levels - 1;
      float* dst = (last && z_out) ? z_out : zbuf[l & 1];
      StageTimer t(c, st, 1 + l);
      HIP_TRY(run_blur_level(c->levels[l], src ? src + b0 * npix : nullptr, hm + b0 * npix, c->gel_dev, sa + b0, sb + b0, pd + b0,
                             dst + b0 * npix, tmp, last && mask_out ? mask_out + b0 * npix : nullptr, nb,
                             c->H, c->W, c->contact_scale, last ? 0 : 1, l == 0, st, rows ? rows + 4 * b0 : nullptr, grow, grow_x),
              "blur level");
      grow += (c->levels[l].kh - 1) / 2;
      grow_x += (c->levels[l].kw - 1) / 2;
      src = dst;
    }
  }
  if (dual) {  // the tail (and whatever the caller enqueues next) waits for the odd chunks
    for (int q = 0; q < lvl_streams - 1; ++q) {
      HIP_TRY(hipEventRecord(c->lvl_join[q], c->lvl_stream[q]), "hipEventRecord");
      HIP_TRY(hipStreamWaitEvent(st, c->lvl_join[q], 0), "hipStreamWaitEvent");
    }
  }
  int band_grow = 0;  // rows by which the band levels spread the non-zero range of their output beyond the contact rows
  for (int l = 0; l < n_band; ++l) band_grow += (c->levels[l].kh - 1) / 2;
  if (stream_tail) {
    // trailing small-kernel levels (+ restores), shading, observation and FOTS by-products: wave-autonomous streaming kernel
    StageTimer t(c, st, c->n_levels + 2);
    const bool want_obs = obs_h && obs;
    const StreamPlan* plan = nullptr;
    if (int rc = stream_plan(c, n_fused, B, want_obs ? obs_hh : 0, want_obs ? obs_w : 0, &plan)) return rc;
    const size_t obs_scratch_floats = (size_t)B * (size_t)(c->H * obs_w > obs_hh * c->W ? c->H * obs_w : obs_hh * c->W) * 3;
    const bool fuse_obs = want_obs && plan->obs_ready &&
                          (size_t)B * plan->nstrips * plan->nseg * plan->obs_nrows * plan->obs_ncols * 3 <= obs_scratch_floats;
    const bool pix = frame0 >= 0 && c->fots_pix_z && c->mk_x;
    float* z_last = src == zbuf[0] ? zbuf[1] : zbuf[0];  // the level buffer the last band level did not write
    const int n_items = B * plan->nstrips * plan->nseg;
    HIP_TRY(run_stream_tail(c->levels, c->n_levels, n_fused, src, hm, c->gel_dev, sa, sb, pd, &c->shade, rgb, z_last, B, c->H, c->W,
                            c->contact_scale, *plan, fuse_obs ? obs_h : nullptr, fots_part,
                            (int)(tail_tiles_per_frame(c->H, c->W) * kTailWavesPerTile),
                            pix ? c->fots_pix_z + (size_t)frame0 * c->fots_taps.n_markers : nullptr,
                            pix ? c->fots_pix_m + (size_t)frame0 * c->fots_taps.n_markers : nullptr, st, rows, band_grow,
                            n_items <= c->stream_order_cap ? c->stream_order : nullptr, order_done),
            "taxim_stream_kernel");
    if (fuse_obs) {
      HIP_TRY(run_obs_finish_stream(obs_h, obs, obs_u8, *plan, B, st), "obs_finish_stream_kernel");
    } else if (want_obs) {
      if (int rc = resize_obs(c, rgb, obs_h, obs, obs_u8, obs_hh, obs_w, B, st)) return rc;
    }
    return 0;
  }
  if (n_fused > 0) {
    // trailing small-kernel levels (+ restores) and the shading in one LDS-tiled kernel
    StageTimer t(c, st, c->n_levels + 2);
    // fused observation: needs down-sampling factors >= 7.5 (y) / 8 (x) (cell-count bounds of the tile) and scratch room
    bool fuse_obs = false;
    int onry = 0, oncx = 0, oky = 0, okx = 0;
    if (obs_h && obs && rgb) {
      if (int rc = ensure_obs_tables(c, obs_hh, obs_w)) return rc;
      const int k0 = c->levels[c->n_levels - n_fused].kw;
      fuse_obs = obs_fusable(c->obs_tab, c->H, c->W, n_fused, k0) && tail_obs_geom(n_fused, k0, &onry, &oncx, &oky, &okx) &&
                 obs_part_floats(c->H, c->W, B, onry, oncx) <= (size_t)B * (size_t)(c->H * obs_w > obs_hh * c->W ? c->H * obs_w : obs_hh * c->W) * 3;
    }
    HIP_TRY(run_tail(c->levels, c->n_levels, n_fused, src, hm, c->gel_dev, sa, sb, pd, z_out, mask_out, &c->shade, rgb,
                     fuse_obs ? obs_h : nullptr, fuse_obs ? &c->obs_tab : nullptr, fots_part,
                     (frame0 >= 0 && c->fots_pix_z) ? &c->fots_taps : nullptr,
                     (frame0 >= 0 && c->fots_pix_z) ? c->fots_pix_z + (size_t)frame0 * c->fots_taps.n_markers : nullptr,
                     (frame0 >= 0 && c->fots_pix_m) ? c->fots_pix_m + (size_t)frame0 * c->fots_taps.n_markers : nullptr, B, c->H, c->W,
                     c->contact_scale, st),
            "taxim_tail_kernel");
    if (fuse_obs) {
      HIP_TRY(run_obs_finish(obs_h, obs, obs_u8, c->obs_tab, c->H, c->W, B, onry, oncx, st), "obs_finish_kernel");
    } else if (obs && rgb) {  // no fusable geometry: plain two-pass down-sample of the finished frame (obs_h = scratch)
      if (int rc = resize_obs(c, rgb, obs_h, obs, obs_u8, obs_hh, obs_w, B, st)) return rc;
    }
    return 0;
  }
  if (rgb) {
    StageTimer t(c, st, c->n_levels + 1);
    HIP_TRY(run_shade(c->shade, src, rgb, nullptr, B, st), "shade_kernel");
  }
  if (obs && rgb) return resize_obs(c, rgb, obs_h, obs, obs_u8, obs_hh, obs_w, B, st);
  return 0;
}

int tacex_taxim_deform(tacex_taxim_ctx* c, const float* hm, const float* press, float* frame_min,
                       float* z_out, uint8_t* mask_out, void* ws, int B, unsigned flags, void* stream) {
  if (!c || !hm || !frame_min || !z_out || !ws) { set_error("tacex_taxim_deform: null argument"); return 2; }
  if (!press && !(flags & TACEX_FLAG_NO_SHIFT)) { set_error("tacex_taxim_deform: press_dev is null"); return 2; }
  if (B <= 0) return 0;
  return pipeline_impl(c, hm, press, frame_min, nullptr, z_out, mask_out, ws, B, flags, (hipStream_t)stream);
}

int tacex_taxim_shade(tacex_taxim_ctx* c, const float* z, float* rgb, uint8_t* idx_out, int B, void* stream) {
  if (!c || !z || !rgb) { set_error("tacex_taxim_shade: null argument"); return 2; }
  if (B <= 0) return 0;
  StageTimer t(c, (hipStream_t)stream, c->n_levels + 1);
  HIP_TRY(run_shade(c->shade, z, rgb, idx_out, B, (hipStream_t)stream), "shade_kernel");
  return 0;
}

int tacex_taxim_render(tacex_taxim_ctx* c, const float* hm, const float* press, float* frame_min, float* rgb,
                       float* z_out, uint8_t* mask_out, void* ws, int B, unsigned flags, void* stream) {
  if (!c || !hm || !frame_min || !rgb || !ws) { set_error("tacex_taxim_render: null argument"); return 2; }
  if (!press && !(flags & TACEX_FLAG_NO_SHIFT)) { set_error("tacex_taxim_render: press_dev is null"); return 2; }
  if (B <= 0) return 0;
  if (flags & TACEX_FLAG_WITH_SHADOW) {
    if (!c->shadow.ready) { set_error("tacex_taxim_render: TACEX_FLAG_WITH_SHADOW needs tacex_taxim_set_shadow first"); return 2; }
    const size_t img = align_up((size_t)B * c->H * c->W * sizeof(float), 256);
    char* sw = static_cast<char*>(ws) + tacex_taxim_workspace_bytes(c, B);
    float* zb = z_out ? z_out : reinterpret_cast<float*>(sw);
    uint8_t* mb = mask_out ? mask_out : reinterpret_cast<uint8_t*>(sw + img);
    float* gdir = reinterpret_cast<float*>(sw + 2 * img);
    float* raw = reinterpret_cast<float*>(sw + 3 * img);
    float* shd = reinterpret_cast<float*>(sw + 6 * img);
    float* tmp = reinterpret_cast<float*>(sw + 9 * img);
    int rc = pipeline_impl(c, hm, press, frame_min, nullptr, zb, mb, ws, B, flags, (hipStream_t)stream);
    if (rc) return rc;
    StageTimer t(c, (hipStream_t)stream, c->n_levels + 1);
    HIP_TRY(run_shadow(c->shadow, c->shade, zb, mb, c->gel_dev, rgb, raw, shd, gdir, tmp, B, (hipStream_t)stream), "shadow branch");
    return 0;
  }
  return pipeline_impl(c, hm, press, frame_min, rgb, z_out, mask_out, ws, B, flags, (hipStream_t)stream);
}

int tacex_taxim_set_frame_rows(tacex_taxim_ctx* c, const int32_t* frame_rows, int capacity_frames) {
  if (!c) { set_error("tacex_taxim_set_frame_rows: null context"); return 2; }
  c->frame_rows = (frame_rows && capacity_frames > 0) ? frame_rows : nullptr;
  c->frame_rows_cap = c->frame_rows ? capacity_frames : 0;
  return 0;
}

int tacex_taxim_set_fots_taps(tacex_taxim_ctx* c, const int32_t* marker_x, const int32_t* marker_y, int n_markers,
                              float* z_pix_dev, uint8_t* mask_pix_dev, int capacity_frames) {
  if (!c) { set_error("tacex_taxim_set_fots_taps: null context"); return 2; }
  if (!z_pix_dev || !mask_pix_dev || !marker_x || !marker_y || n_markers <= 0) {  // disable
    c->fots_pix_z = nullptr; c->fots_pix_m = nullptr; c->fots_pix_cap = 0;
    c->mk_row_ptr_h.clear(); c->mk_x_h.clear(); c->mk_id_h.clear(); c->mk_x = nullptr; c->mk_id = nullptr; ++c->mk_version;
    return 0;
  }
  if (n_markers > 65535) { set_error("tacex_taxim_set_fots_taps: too many markers"); return 2; }
  HIP_TRY(hipSetDevice(c->device), "hipSetDevice");
  const int ntx = (c->W + 63) / 64, nty = (c->H + 31) / 32;
  std::vector<int> tile((size_t)ntx * nty * kTailMaxMarkersPerTile, 0), cnt((size_t)ntx * nty, 0);
  for (int m = 0; m < n_markers; ++m) {
    const int x = marker_x[m], y = marker_y[m];
    if (x < 0 || x >= c->W || y < 0 || y >= c->H) continue;  // FOTS ignores markers outside the image (MM:152-166)
    const int t = (y / 32) * ntx + x / 64;
    if (cnt[t] >= kTailMaxMarkersPerTile) { set_error("tacex_taxim_set_fots_taps: more than %d markers in one 64x32 tile", kTailMaxMarkersPerTile); return 2; }
    tile[(size_t)t * kTailMaxMarkersPerTile + cnt[t]++] = (m << 16) | ((y % 32) << 8) | (x % 64);
  }
  int *dt = nullptr, *dc = nullptr;
  if (int rc = upload(c, tile.data(), tile.size(), &dt)) return rc;
  if (int rc = upload(c, cnt.data(), cnt.size(), &dc)) return rc;
  {  // the same markers as a CSR over frame rows (streaming tail)
    std::vector<int> ptr(c->H + 1, 0), mx, mid;
    for (int m = 0; m < n_markers; ++m)
      if (marker_x[m] >= 0 && marker_x[m] < c->W && marker_y[m] >= 0 && marker_y[m] < c->H) ++ptr[marker_y[m] + 1];
    for (int y = 0; y < c->H; ++y) ptr[y + 1] += ptr[y];
    mx.resize(ptr[c->H] > 0 ? ptr[c->H] : 1); mid.resize(mx.size());
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int m = 0; m < n_markers; ++m)
      if (marker_x[m] >= 0 && marker_x[m] < c->W && marker_y[m] >= 0 && marker_y[m] < c->H) {
        const int e = fill[marker_y[m]]++;
        mx[e] = marker_x[m]; mid[e] = m;
      }
    int *dx = nullptr, *di = nullptr;
    if (int rc = upload(c, mx.data(), mx.size(), &dx)) return rc;
    if (int rc = upload(c, mid.data(), mid.size(), &di)) return rc;
    c->mk_row_ptr_h = ptr; c->mk_x_h = mx; c->mk_id_h = mid; c->mk_x = dx; c->mk_id = di; ++c->mk_version;
  }
  c->fots_taps.mk_tile = dt; c->fots_taps.mk_cnt = dc; c->fots_taps.n_markers = n_markers;
  c->fots_pix_z = z_pix_dev; c->fots_pix_m = mask_pix_dev; c->fots_pix_cap = capacity_frames;
  return 0;
}

int tacex_taxim_fots_partials_per_env(const tacex_taxim_ctx* c) {
  if (!c || !c->use_tail || c->n_fused <= 0) return 0;
  return (int)(tail_tiles_per_frame(c->H, c->W) * kTailWavesPerTile);
}

int tacex_taxim_set_fots_partials(tacex_taxim_ctx* c, void* partials_dev, int capacity_frames) {
  if (!c) { set_error("tacex_taxim_set_fots_partials: null context"); return 2; }
  c->fots_part = static_cast<FotsReduce*>(partials_dev);
  c->fots_cap = partials_dev ? capacity_frames : 0;
  return 0;
}

/* render + low-resolution policy observation in the same pass */
int tacex_taxim_render_obs(tacex_taxim_ctx* c, const float* hm, const float* press, float* frame_min, float* rgb,
                           float* z_out, uint8_t* mask_out, void* ws, float* obs_scratch, void* obs_out, int obs_h,
                           int obs_w, int B, unsigned flags, void* stream) {
  if (!c || !hm || !frame_min || !rgb || !ws || !obs_scratch || !obs_out) { set_error("tacex_taxim_render_obs: null argument"); return 2; }
  if (obs_h <= 0 || obs_w <= 0 || obs_h > c->H || obs_w > c->W) { set_error("tacex_taxim_render_obs: bad observation size %dx%d", obs_w, obs_h); return 2; }
  if (!press && !(flags & TACEX_FLAG_NO_SHIFT)) { set_error("tacex_taxim_render_obs: press_dev is null"); return 2; }
  if (B <= 0) return 0;
  if (flags & TACEX_FLAG_WITH_SHADOW) {  // the shadow branch re-blurs the finished frame: its observation is a plain two-pass resize
    if (int rc = tacex_taxim_render(c, hm, press, frame_min, rgb, z_out, mask_out, ws, B, flags & ~TACEX_FLAG_OBS_U8, stream)) return rc;
    return resize_obs(c, rgb, obs_scratch, obs_out, (flags & TACEX_FLAG_OBS_U8) != 0, obs_h, obs_w, B, (hipStream_t)stream);
  }
  return pipeline_impl(c, hm, press, frame_min, rgb, z_out, mask_out, ws, B, flags, (hipStream_t)stream, obs_scratch, obs_out,
                       obs_h, obs_w);
}

/* test / ablation hook: 0 = run every level as its own kernel + separate shade, 1 = fused tail (default) */
int tacex_taxim_set_fused_tail(tacex_taxim_ctx* c, int enabled) {
  if (!c) { set_error("null ctx"); return 2; }
  c->use_tail = enabled != 0;
  c->use_stream = enabled == 1;
  return 0;
}

int tacex_resize_bilinear_aa(const float* src, int sh, int sw, float* dst, int dh, int dw, int B, void* stream) {
  if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) { set_error("tacex_resize_bilinear_aa: bad argument"); return 2; }
  if (B <= 0) return 0;
  HIP_TRY(run_resize_aa(src, sh, sw, dst, dh, dw, B, 1, nullptr, (hipStream_t)stream), "resize_aa_kernel");
  return 0;
}

int tacex_resize_bilinear_aa_nhwc(const float* src, int sh, int sw, float* dst, int dh, int dw, int channels, int B,
                                  float* tmp, void* stream) {
  if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || channels <= 0) { set_error("tacex_resize_bilinear_aa_nhwc: bad argument"); return 2; }
  if (B <= 0) return 0;
  HIP_TRY(run_resize_aa(src, sh, sw, dst, dh, dw, B, channels, tmp, (hipStream_t)stream), "resize_aa_kernel");
  return 0;
}

}  // extern "C"
