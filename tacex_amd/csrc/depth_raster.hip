// Height-map SOURCE for arbitrary rigid indenters (second slice): pinhole depth image of a triangle mesh per
// env - what the IsaacLab TiledCamera hands GelSightSensor._get_height_map (GS:229-263, 581-593: "distance_to_image_plane"
// depth in metres, inf where the camera sees nothing inside its clipping range).  One shared mesh (object frame), one rigid pose
// per env (unit quaternion wxyz + translation into the camera frame: x right, y down, z along the optical axis).
//
// Workgroup = (env, 64 x 32 pixel tile) with the tile's z-buffer in LDS; every thread walks the triangles t = tid, tid + 256, ...:
// transform, project, clip the bounding box to the tile, and for the covered pixel centres (j + 0.5, i + 0.5) interpolate 1/z
// (affine in screen space for a pinhole) and atomicMin the depth (positive floats order like their bit patterns).  No back-face
// culling (the nearest surface wins whatever its orientation), fragments outside [near, far] are dropped per pixel (clipping),
// triangles with a vertex at or behind the camera plane are dropped whole.  Arithmetic is plain float32 with FMA contraction
// off (this file is compiled with -ffp-contract=off, tacex_amd/_build.py) and IEEE division - see oracle/mesh_depth_oracle.py, which repeats them in the same order: the two agree bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "tacex_hip.h"
#include "tacex_internal.h"

namespace tacex {

constexpr int kRasterTileW = 64, kRasterTileH = 32;

struct RasterArgs {
  const float* verts;   // (V,3)
  const int* tris;      // (T,3)
  const float* pos;     // (B,3) translation into the camera frame
  const float* quat;    // (B,4) wxyz rotation into the camera frame (normalised here, in double)
  float* depth;         // (B,H,W)
  int V, T, B, H, W;
  float fx, fy, cx, cy, near_m, far_m;
  int tiles_x, tiles_y;
  float bs[4];          // bounding sphere of the mesh in the object frame (centre, radius); radius < 0: unknown
};

__global__ __launch_bounds__(256) void mesh_depth_kernel(RasterArgs a) {
  __shared__ unsigned zbuf[kRasterTileW * kRasterTileH];
  const int per_env = a.tiles_x * a.tiles_y;
  const int env = blockIdx.x / per_env, tile = blockIdx.x - env * per_env;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int x0 = tx * kRasterTileW, y0 = ty * kRasterTileH;
  const int x1 = min(x0 + kRasterTileW, a.W), y1 = min(y0 + kRasterTileH, a.H);  // exclusive
  for (int i = threadIdx.x; i < kRasterTileW * kRasterTileH; i += blockDim.x) zbuf[i] = 0x7f800000u;  // +inf
  __syncthreads();
  // rotation matrix of the env's quaternion: double arithmetic, rounded once to float32 (what the NumPy restatement does)
  double qw = a.quat[4 * env], qx = a.quat[4 * env + 1], qy = a.quat[4 * env + 2], qz = a.quat[4 * env + 3];
  const double qn = sqrt(((qw * qw + qx * qx) + qy * qy) + qz * qz);
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  const float r00 = (float)(1.0 - 2.0 * (qy * qy + qz * qz)), r01 = (float)(2.0 * (qx * qy - qz * qw)), r02 = (float)(2.0 * (qx * qz + qy * qw));
  const float r10 = (float)(2.0 * (qx * qy + qz * qw)), r11 = (float)(1.0 - 2.0 * (qx * qx + qz * qz)), r12 = (float)(2.0 * (qy * qz - qx * qw));
  const float r20 = (float)(2.0 * (qx * qz - qy * qw)), r21 = (float)(2.0 * (qy * qz + qx * qw)), r22 = (float)(1.0 - 2.0 * (qx * qx + qy * qy));
  const float t0 = a.pos[3 * env], t1 = a.pos[3 * env + 1], t2 = a.pos[3 * env + 2];
  // Tiles the mesh cannot touch skip the triangle loop (a contact covers a few percent of the image): conservative screen
  // bounds of the mesh's bounding sphere - x, y in [c -+ r] over z in [c.z - r, c.z + r] - against the tile.  Exactness is not
  // at stake: a skipped tile holds no fragment.
  bool tile_empty = false;
  if (a.bs[3] >= 0.0f) {
    const float r = a.bs[3];
    const float bx = ((r00 * a.bs[0] + r01 * a.bs[1]) + r02 * a.bs[2]) + t0;
    const float by = ((r10 * a.bs[0] + r11 * a.bs[1]) + r12 * a.bs[2]) + t1;
    const float bz = ((r20 * a.bs[0] + r21 * a.bs[1]) + r22 * a.bs[2]) + t2;
    const float zn = bz - r, zf = bz + r;
    if (zf < a.near_m || zn > a.far_m) {
      tile_empty = true;
    } else if (zn > 1e-4f) {
      const float m = 1.001f;  // slack for the rounding of the bounds themselves
      const float ulo = fminf(a.fx * (bx - r * m) / zn, a.fx * (bx - r * m) / zf) + a.cx - 1.0f;
      const float uhi = fmaxf(a.fx * (bx + r * m) / zn, a.fx * (bx + r * m) / zf) + a.cx + 1.0f;
      const float vlo = fminf(a.fy * (by - r * m) / zn, a.fy * (by - r * m) / zf) + a.cy - 1.0f;
      const float vhi = fmaxf(a.fy * (by + r * m) / zn, a.fy * (by + r * m) / zf) + a.cy + 1.0f;
      tile_empty = uhi < (float)x0 || ulo > (float)x1 || vhi < (float)y0 || vlo > (float)y1;
    }
  }
  for (int t = tile_empty ? a.T : threadIdx.x; t < a.T; t += blockDim.x) {
    float sx[3], sy[3], iz[3];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int vi = a.tris[3 * t + k];
      const float vx = a.verts[3 * vi], vy = a.verts[3 * vi + 1], vz = a.verts[3 * vi + 2];
      const float px = ((r00 * vx + r01 * vy) + r02 * vz) + t0;
      const float py = ((r10 * vx + r11 * vy) + r12 * vz) + t1;
      const float pz = ((r20 * vx + r21 * vy) + r22 * vz) + t2;
      ok = ok && pz > 1e-6f;
      iz[k] = 1.0f / pz;
      sx[k] = (a.fx * px) * iz[k] + a.cx;
      sy[k] = (a.fy * py) * iz[k] + a.cy;
    }
    if (!ok) continue;
    const float minx = fminf(fminf(sx[0], sx[1]), sx[2]), maxx = fmaxf(fmaxf(sx[0], sx[1]), sx[2]);
    const float miny = fminf(fminf(sy[0], sy[1]), sy[2]), maxy = fmaxf(fmaxf(sy[0], sy[1]), sy[2]);
    // pixel centres j + 0.5 inside [minx, maxx]: j from ceil(minx - 0.5) to floor(maxx - 0.5)
    const int jx0 = max(x0, (int)ceilf(minx - 0.5f)), jx1 = min(x1 - 1, (int)floorf(maxx - 0.5f));
    const int iy0 = max(y0, (int)ceilf(miny - 0.5f)), iy1 = min(y1 - 1, (int)floorf(maxy - 0.5f));
    if (jx0 > jx1 || iy0 > iy1) continue;
    const float area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sy[1] - sy[0]) * (sx[2] - sx[0]);
    if (area == 0.0f) continue;
    const float inv_area = 1.0f / area;
    for (int i = iy0; i <= iy1; ++i) {
      const float py = (float)i + 0.5f;
      for (int j = jx0; j <= jx1; ++j) {
        const float px = (float)j + 0.5f;
        // edge functions (twice the signed sub-triangle areas); inside when all share the sign of `area` (zero counts as inside)
        const float e0 = (sx[2] - sx[1]) * (py - sy[1]) - (sy[2] - sy[1]) * (px - sx[1]);
        const float e1 = (sx[0] - sx[2]) * (py - sy[2]) - (sy[0] - sy[2]) * (px - sx[2]);
        const float e2 = (sx[1] - sx[0]) * (py - sy[0]) - (sy[1] - sy[0]) * (px - sx[0]);
        const bool in = area > 0.0f ? (e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f) : (e0 <= 0.0f && e1 <= 0.0f && e2 <= 0.0f);
        if (!in) continue;
        const float l0 = e0 * inv_area, l1 = e1 * inv_area, l2 = e2 * inv_area;
        const float invz = (l0 * iz[0] + l1 * iz[1]) + l2 * iz[2];
        const float z = 1.0f / invz;
        if (!(z >= a.near_m && z <= a.far_m)) continue;  // clipping range of the camera (also drops NaN)
        atomicMin(&zbuf[(i - y0) * kRasterTileW + (j - x0)], __float_as_uint(z));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRasterTileW * kRasterTileH; i += blockDim.x) {
    const int yy = y0 + i / kRasterTileW, xx = x0 + i % kRasterTileW;
    if (yy < y1 && xx < x1) a.depth[((size_t)env * a.H + yy) * a.W + xx] = __uint_as_float(zbuf[i]);
  }
}

}  // namespace tacex

extern "C" int tacex_depth_from_mesh(const float* verts_dev, const int32_t* tris_dev, int num_verts, int num_tris,
                                     const float* pos_dev, const float* quat_dev, float fx, float fy, float cx, float cy, float near_clip_m,
                                     float far_clip_m, const float* bounding_sphere, float* depth_m_dev, int num_envs, int height,
                                     int width, void* stream) {
  using namespace tacex;
  if (!verts_dev || !tris_dev || !pos_dev || !quat_dev || !depth_m_dev) { set_error("tacex_depth_from_mesh: null buffer"); return 2; }
  if (num_verts <= 0 || num_tris <= 0 || height <= 0 || width <= 0) { set_error("tacex_depth_from_mesh: empty mesh or image"); return 2; }
  if (!(near_clip_m >= 0.0f) || !(far_clip_m > near_clip_m)) { set_error("tacex_depth_from_mesh: clipping range (%g, %g)", near_clip_m, far_clip_m); return 2; }
  if (num_envs <= 0) return 0;
  RasterArgs a{};
  a.verts = verts_dev; a.tris = tris_dev; a.pos = pos_dev; a.quat = quat_dev; a.depth = depth_m_dev;
  a.V = num_verts; a.T = num_tris; a.B = num_envs; a.H = height; a.W = width;
  a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.near_m = near_clip_m; a.far_m = far_clip_m;
  if (bounding_sphere) { a.bs[0] = bounding_sphere[0]; a.bs[1] = bounding_sphere[1]; a.bs[2] = bounding_sphere[2]; a.bs[3] = bounding_sphere[3]; }
  else a.bs[3] = -1.0f;
  a.tiles_x = (width + kRasterTileW - 1) / kRasterTileW; a.tiles_y = (height + kRasterTileH - 1) / kRasterTileH;
  hipLaunchKernelGGL(mesh_depth_kernel, dim3((unsigned)(num_envs * a.tiles_x * a.tiles_y)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("mesh_depth_kernel: %s", hipGetErrorString(e)); return 1; }
  return 0;
}
