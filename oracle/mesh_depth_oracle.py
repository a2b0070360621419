"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of the mesh depth source
(tacex_amd/csrc/depth_raster.hip) - the pinhole "distance_to_image_plane" depth image an IsaacLab TiledCamera
hands GelSightSensor._get_height_map (reference gelsight_sensor.py:229-263, 581-593).

PARITY UNPINNED vs the reference: its depth comes from Isaac Sim's renderer, which is not in the reference tree.  The oracle is
pinned instead by a closed form (tests/test_mesh_depth.py: a finely tessellated sphere against the analytic ray / sphere
depth) and the HIP kernel is compared with it bit for bit (same float32 operations in the same order, no FMA).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def quat_to_matrix(q_wxyz: np.ndarray) -> np.ndarray:
    q = np.asarray(q_wxyz, dtype=np.float64)
    w, x, y, z = (q / np.sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3])).tolist()  # float64, the kernel's order
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_rows(pos: np.ndarray, quat_wxyz: np.ndarray) -> np.ndarray:
    """(B,3) positions + (B,4) quaternions -> (B,12) float32 [R row-major | t]: the matrix the kernel builds from the quaternion
    (float64 arithmetic, rounded once)."""
    out = np.zeros((len(pos), 12), dtype=np.float32)
    for b in range(len(pos)):
        out[b, :9] = quat_to_matrix(quat_wxyz[b]).reshape(-1)
        out[b, 9:] = pos[b]
    return out


def render_depth(verts, tris, pose12, fx, fy, cx, cy, near, far, H, W) -> np.ndarray:
    """(B,H,W) float32 depth [m], inf where no fragment lies inside [near, far].  float32 throughout, operation order of the kernel."""
    verts = np.asarray(verts, dtype=F)
    tris = np.asarray(tris, dtype=np.int64)
    fx, fy, cx, cy, near, far = F(fx), F(fy), F(cx), F(cy), F(near), F(far)
    B = pose12.shape[0]
    depth = np.full((B, H, W), np.inf, dtype=F)
    half = F(0.5)
    for b in range(B):
        P = pose12[b].astype(F)
        R, t = P[:9].reshape(3, 3), P[9:]
        v = verts
        px = ((R[0, 0] * v[:, 0] + R[0, 1] * v[:, 1]) + R[0, 2] * v[:, 2]) + t[0]
        py = ((R[1, 0] * v[:, 0] + R[1, 1] * v[:, 1]) + R[1, 2] * v[:, 2]) + t[1]
        pz = ((R[2, 0] * v[:, 0] + R[2, 1] * v[:, 1]) + R[2, 2] * v[:, 2]) + t[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            iz = (F(1.0) / pz).astype(F)
            sx = ((fx * px) * iz + cx).astype(F)
            sy = ((fy * py) * iz + cy).astype(F)
        front = pz > F(1e-6)
        for tri in tris:
            if not front[tri].all():
                continue
            x, y, z = sx[tri], sy[tri], iz[tri]
            j0 = max(0, int(np.ceil(x.min() - half))); j1 = min(W - 1, int(np.floor(x.max() - half)))
            i0 = max(0, int(np.ceil(y.min() - half))); i1 = min(H - 1, int(np.floor(y.max() - half)))
            if j0 > j1 or i0 > i1:
                continue
            area = F((x[1] - x[0]) * (y[2] - y[0]) - (y[1] - y[0]) * (x[2] - x[0]))
            if area == 0:
                continue
            inv_area = F(1.0) / area
            qy = (np.arange(i0, i1 + 1, dtype=F) + half)[:, None]
            qx = (np.arange(j0, j1 + 1, dtype=F) + half)[None, :]
            e0 = ((x[2] - x[1]) * (qy - y[1]) - (y[2] - y[1]) * (qx - x[1])).astype(F)
            e1 = ((x[0] - x[2]) * (qy - y[2]) - (y[0] - y[2]) * (qx - x[2])).astype(F)
            e2 = ((x[1] - x[0]) * (qy - y[0]) - (y[1] - y[0]) * (qx - x[0])).astype(F)
            inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) if area > 0 else ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
            if not inside.any():
                continue
            l0, l1, l2 = (e0 * inv_area).astype(F), (e1 * inv_area).astype(F), (e2 * inv_area).astype(F)
            with np.errstate(divide="ignore", invalid="ignore"):
                zz = (F(1.0) / ((l0 * z[0] + l1 * z[1]) + l2 * z[2]).astype(F)).astype(F)
            ok = inside & (zz >= near) & (zz <= far)
            blk = depth[b, i0:i1 + 1, j0:j1 + 1]
            np.minimum(blk, np.where(ok, zz, np.inf).astype(F), out=blk)
    return depth


def icosphere(radius: float, subdivisions: int = 3):
    """Unit icosahedron subdivided `subdivisions` times, scaled to `radius`: (V,3) float32, (T,3) int32."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius).astype(np.float32), np.array(f, dtype=np.int32)
