"""`VisionTactileSensorUIPC` - FEM-driven marker flow on the gelpad surface, counterpart of
source/tacex/tacex/simulation_approaches/fem_based/sim/tactile_sensor_sapienipc_modified.py:42-413.

What the reference does every call on the host (NumPy + sklearn + usdrt, single env, VT:147-154 "todo fix it for multi
env"): build the marker grid (VT:189-247), find for every marker the surface triangle under it and its barycentric
weights (VT:249-329, `in_hull` geometry.py:86-100), interpolate reference and current surface vertices (VT:359-366),
project with a pinhole camera (VT:331-352) and mask / pad to `num_markers` (VT:382-405).

Here the grid + weights are host-side set-up (computed once while the random ranges are degenerate - the default - and
re-drawn per call otherwise, like the reference), and the per-step part - barycentric points + projection for ALL envs -
is one HIP launch (`tacex_fem_marker_uv`).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .... import _lib


def in_hull(p: np.ndarray, hull_pts: np.ndarray) -> np.ndarray:
    """Points of `p` (N,K) inside the convex hull of `hull_pts` (M,K) (geometry.py:86-100)."""
    from scipy.spatial import Delaunay

    return Delaunay(hull_pts).find_simplex(p) >= 0


def gen_marker_grid(marker_interval_range=(2.0625, 2.0625), marker_rotation_range=0.0,
                    marker_translation_range=(0.0, 0.0), marker_pos_shift_range=(0.0, 0.0), rng=None) -> np.ndarray:
    """(M,2) marker positions in metres over x in [-8, 16.5] mm, y in [-6, 6] mm (VT:189-247, same draw order)."""
    rng = np.random if rng is None else rng
    interval = (marker_interval_range[1] - marker_interval_range[0]) * rng.rand(1)[0] + marker_interval_range[0]
    rot = float((2 * marker_rotation_range * rng.rand(1) - marker_rotation_range)[0])
    tx = 2 * marker_translation_range[0] * rng.rand(1)[0] - marker_translation_range[0]
    ty = 2 * marker_translation_range[1] * rng.rand(1)[0] - marker_translation_range[1]
    x_start = -math.ceil((8 + tx) / interval) * interval + tx
    x_end = math.ceil((16.5 - tx) / interval) * interval + tx
    y_start = -math.ceil((6 + ty) / interval) * interval + ty
    y_end = math.ceil((6 - ty) / interval) * interval + ty
    mx = np.linspace(x_start, x_end, round((x_end - x_start) / interval) + 1, True)
    my = np.linspace(y_start, y_end, round((y_end - y_start) / interval) + 1, True)
    xy = np.array(np.meshgrid(mx, my)).reshape((2, -1)).T
    n = xy.shape[0]
    xy[:, 0] += rng.rand(n) * marker_pos_shift_range[0] * 2 - marker_pos_shift_range[0]
    xy[:, 1] += rng.rand(n) * marker_pos_shift_range[1] * 2 - marker_pos_shift_range[1]
    rot_mat = np.array([[math.cos(rot), -math.sin(rot)], [math.sin(rot), math.cos(rot)]])
    return (xy @ rot_mat.T) / 1000.0


def gen_marker_weight(marker_pts_xy: np.ndarray, surface_pts: np.ndarray, triangles: np.ndarray):
    """Triangle vertex ids (M',3) + barycentric weights (M',3) for the markers that lie over the surface (VT:249-329).
    surface_pts (Vs,3) camera frame, triangles (F,3) indices into surface_pts."""
    from scipy.spatial import cKDTree

    z = np.max(surface_pts[:, 2])  # pattern on the far side of the gelpad (VT:262-265)
    pts = np.hstack((marker_pts_xy, np.ones((marker_pts_xy.shape[0], 1)) * z))
    pts = pts[in_hull(pts[:, :2], surface_pts[:, :2])]
    f_centers = surface_pts[triangles].mean(axis=1)
    k = min(4, len(f_centers))
    _, face_idx = cKDTree(f_centers).query(pts, k=k)  # reference: sklearn NearestNeighbors(4, ball_tree)
    face_idx = face_idx.reshape(len(pts), k)
    p2 = pts[:, :2]
    idx, wgt = [], []
    for i in range(p2.shape[0]):
        # faces seen edge-on from the camera (side walls of the pad among the 4 nearest) have no barycentric coordinates in
        # the image plane: the reference's np.linalg.inv raises there (VT:300-303); they are skipped instead
        cand = [f for f in face_idx[i].tolist()
                if abs(np.linalg.det(np.stack([surface_pts[triangles[f]][1, :2] - surface_pts[triangles[f]][0, :2],
                                               surface_pts[triangles[f]][2, :2] - surface_pts[triangles[f]][0, :2]], axis=1))) > 1e-18]
        if not cand:
            raise RuntimeError(f"marker {i}: all nearest surface faces are edge-on to the camera")
        for fid in cand:
            p0, p1, q2 = surface_pts[triangles[fid]][:, :2]
            A = np.stack([p1 - p0, q2 - p0], axis=1)
            w12 = np.linalg.inv(A) @ (p2[i] - p0)
            inside = w12[0] >= 0 and w12[1] >= 0 and w12[0] + w12[1] <= 1
            if fid == cand[0]:
                idx.append(triangles[fid])
                wgt.append(np.array([1 - w12.sum(), w12[0], w12[1]]))
                if inside:
                    break
            elif inside:
                idx[-1] = triangles[fid]
                wgt[-1] = np.array([1 - w12.sum(), w12[0], w12[1]])
                break
    idx = np.stack(idx).astype(np.int32)
    wgt = np.stack(wgt).astype(np.float64)
    rec = (surface_pts[idx] * wgt[..., None]).sum(1)[:, :2]
    assert np.allclose(rec, p2), f"max err: {np.abs(rec - p2).max()}"
    return idx, wgt


def quat_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """(…,4) wxyz unit quaternion -> (…,3,3)."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))


class VisionTactileSensorUIPC:
    def __init__(self, uipc_gelpad, uipc_sim, cam_pos_w: torch.Tensor, cam_quat_w_ros: torch.Tensor,
                 tactile_img_width=320, tactile_img_height=240, marker_interval_range=(2.0625, 2.0625),
                 marker_rotation_range=0.0, marker_translation_range=(0.0, 0.0), marker_pos_shift_range=(0.0, 0.0),
                 marker_random_noise=0.0, marker_lose_tracking_probability=0.0, normalize=False, num_markers=128,
                 camera_params=(340, 325, 160, 125, 0.0), seed: int = 0, **kwargs):
        self.gelpad_obj = uipc_gelpad
        self.uipc_sim = uipc_sim
        self.tactile_img_width, self.tactile_img_height = tactile_img_width, tactile_img_height
        self.marker_interval_range = marker_interval_range
        self.marker_rotation_range = marker_rotation_range
        self.marker_translation_range = marker_translation_range
        self.marker_pos_shift_range = marker_pos_shift_range
        self.marker_random_noise = marker_random_noise
        self.marker_lose_tracking_probability = marker_lose_tracking_probability
        self.normalize = normalize
        self.num_markers = num_markers
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in camera_params[:4])
        self._rng = np.random.RandomState(seed)
        dev = uipc_sim.device
        self.device = dev
        B = uipc_sim.num_envs
        self.cam_pos_w = cam_pos_w.to(dev, torch.float64).reshape(-1, 3).expand(B, 3).contiguous()
        self.cam_rot_inv = quat_to_matrix(cam_quat_w_ros.to(dev, torch.float64).reshape(-1, 4)).transpose(-1, -2).expand(B, 3, 3).contiguous()
        # surface: boundary triangles of the tet mesh re-indexed onto the surface vertex list
        tri_global = uipc_gelpad.surface_triangles()
        self.surf_vertex_ids = np.unique(tri_global.reshape(-1))
        remap = -np.ones(uipc_gelpad.num_verts, dtype=np.int64)
        remap[self.surf_vertex_ids] = np.arange(len(self.surf_vertex_ids))
        self.surf_triangles = remap[tri_global].astype(np.int32)
        self._surf_ids_dev = torch.from_numpy(self.surf_vertex_ids).to(dev)
        self._surf_ids64 = self._surf_ids_dev.to(torch.int64).contiguous()  # (tacex_fem_marker_flow reads the FEM state through them)
        self.init_surface_vertices_camera = self.get_surface_vertices_camera().clone()
        self.reference_surface_vertices_camera = self.init_surface_vertices_camera.clone()
        self._static = (marker_interval_range[0] == marker_interval_range[1] and marker_rotation_range == 0.0
                        and tuple(marker_translation_range) == (0.0, 0.0) and tuple(marker_pos_shift_range) == (0.0, 0.0)
                        and marker_random_noise == 0.0 and marker_lose_tracking_probability == 0.0)
        self._cached = None
        self._static_flow = None
        self._lib = _lib.load_library()

    # -- frames (VT:142-187) -----------------------------------------------------------------------------
    def get_surface_vertices_world(self) -> torch.Tensor:
        self.uipc_sim.wait_for_step()  # (a step enqueued on a side stream: UipcSim.step_done)
        return self.uipc_sim.x[:, self._surf_ids_dev]  # (B,Vs,3) float64

    def transform_world_to_camera_frame(self, v: torch.Tensor) -> torch.Tensor:
        return torch.matmul(v - self.cam_pos_w[:, None, :], self.cam_rot_inv.transpose(-1, -2))

    def get_surface_vertices_camera(self) -> torch.Tensor:
        return self.transform_world_to_camera_frame(self.get_surface_vertices_world()).contiguous()

    def set_reference_surface_vertices_camera(self):
        self.reference_surface_vertices_camera = self.get_surface_vertices_camera().clone()

    @property
    def reference_surface_vertices_camera(self) -> torch.Tensor:
        return self._ref_surface

    @reference_surface_vertices_camera.setter
    def reference_surface_vertices_camera(self, v: torch.Tensor):
        self._ref_surface = v
        self._ref_version = getattr(self, "_ref_version", 0) + 1  # invalidates the cached initial projection / mask

    # -- per-call set-up, cached while nothing is random --------------------------------------------------------
    def _setup(self):
        if self._static and self._cached is not None:
            return self._cached
        grid = gen_marker_grid(self.marker_interval_range, self.marker_rotation_range, self.marker_translation_range,
                               self.marker_pos_shift_range, self._rng)
        surf0 = self.init_surface_vertices_camera[0].cpu().numpy()  # same mesh in every env
        idx, wgt = gen_marker_weight(grid, surf0, self.surf_triangles)
        out = (torch.from_numpy(idx).to(self.device), torch.from_numpy(wgt).to(self.device))
        if self._static:
            self._cached = out
        return out

    def _project(self, surf_cam: torch.Tensor, tri: torch.Tensor, wgt: torch.Tensor) -> torch.Tensor:
        B, Vs, M = surf_cam.shape[0], surf_cam.shape[1], tri.shape[0]
        uv = torch.empty((B, M, 2), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_marker_uv(_lib.ptr(surf_cam), _lib.ptr(tri), _lib.ptr(wgt), self.fx, self.fy, self.cx,
                                               self.cy, _lib.ptr(uv), B, Vs, M, _lib.current_stream_handle(self.device))
        _lib.check(rc, "tacex_fem_marker_uv")
        return uv

    def _gen_marker_flow_static(self, tri, wgt, curr_uv) -> torch.Tensor:
        """gen_marker_flow when nothing about the marker grid is random (the shipped cfgs): the initial projection and the in-image
        mask (VT:382-387) depend only on the reference surface, so they are computed ONCE per reference surface (the boolean-mask
        indexing costs a device sync) and every call is projection + one gather; the per-call random subset (VT:394-399) is drawn
        on the host and travels through a small ring of pinned buffers - no host/device synchronisation per step."""
        init_uv, idx_dev, idx_host = self._static_tables(tri, wgt)
        n = idx_host.size
        if n >= self.num_markers:
            sel = self._draw_subset(idx_host)
        elif n > 0:  # pad by repeating the last marker (VT:400-405)
            sel = torch.cat([idx_dev, idx_dev[-1:].expand(self.num_markers - n)])
        else:
            self.curr_marker_uv = curr_uv
            return torch.zeros((curr_uv.shape[0], 2, self.num_markers, 2), dtype=curr_uv.dtype, device=self.device)
        ret = torch.stack([init_uv.index_select(1, sel), curr_uv.index_select(1, sel)], dim=1)
        if self.normalize:
            ret = ret / (self.tactile_img_width / 2) - 1.0
        self.curr_marker_uv = curr_uv
        return ret

    def _draw_subset(self, idx_host):
        """This step's random subset of the in-image markers (VT:394-399) as a device tensor: drawn on the host, through a small ring of pinned
        buffers - no host/device synchronisation per step."""
        slot = self._sel_pos % len(self._sel_ring)
        buf = self._sel_ring[slot]
        self._sel_pos += 1
        if self._sel_events[slot] is not None:
            # the host may run more than a ring's worth of steps ahead of the device: overwriting a pinned buffer whose copy has
            # not executed yet would hand an earlier step the wrong subset.  Waits only in that case (eight steps behind).
            self._sel_events[slot].synchronize()
        buf.numpy()[:] = idx_host[self._rng.choice(idx_host.size, self.num_markers, replace=False)]
        sel = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._sel_events[slot] = ev
        return sel

    def _static_tables(self, tri, wgt):
        """(initial projections (B,M,2), in-image marker ids on the device / host) of the current reference surface (VT:382-387)."""
        key = self._ref_version
        if self._static_flow is None or self._static_flow[0] != key:
            init_uv = self._project(self.reference_surface_vertices_camera, tri, wgt)
            u0, v0 = init_uv[0, :, 0], init_uv[0, :, 1]
            mask = (u0 > 5) & (u0 < self.tactile_img_height) & (v0 > 5) & (v0 < self.tactile_img_width)
            idx = torch.nonzero(mask).reshape(-1)
            self._static_flow = (key, init_uv, idx, idx.cpu().numpy())
            self._sel_ring = [torch.empty(self.num_markers, dtype=torch.int64).pin_memory() for _ in range(8)]
            self._sel_events = [None] * len(self._sel_ring)  # recorded behind each slot's H2D copy: a slot is reused only once its copy ran
            self._sel_pos = 0
        return self._static_flow[1:]

    def gen_marker_flow_fused(self, out_f32: torch.Tensor | None = None) -> torch.Tensor | None:
        """gen_marker_flow for a static marker grid with enough in-image markers, as ONE launch on the FEM state (`tacex_fem_marker_flow`):
        returns the (B,2,num_markers,2) float64 flow - or, with `out_f32` (B,2,num_markers,2) float32 contiguous, writes that and returns it.
        None when this path does not apply (random grid, no marker in the image): the caller takes gen_marker_flow()."""
        if not self._static:
            return None
        tri, wgt = self._setup()
        init_uv, idx_dev, idx_host = self._static_tables(tri, wgt)
        if idx_host.size >= self.num_markers:
            sel = self._draw_subset(idx_host)
        elif idx_host.size > 0:  # fewer in-image markers than asked for: all of them, padded by repeating the last one (VT:400-405) - a fixed list
            pad = getattr(self, "_pad_sel", None)
            if pad is None or pad[0] != self._ref_version:
                pad = (self._ref_version, torch.cat([idx_dev, idx_dev[-1:].expand(self.num_markers - idx_host.size)]).contiguous())
                self._pad_sel = pad
            sel = pad[1]
        else:
            return None
        self.uipc_sim.wait_for_step()
        x = self.uipc_sim.x
        B, V, M, K = x.shape[0], x.shape[1], tri.shape[0], self.num_markers
        curr_uv = torch.empty((B, M, 2), dtype=torch.float64, device=self.device)  # (a fresh tensor per call, like the general path: callers may keep the last one)
        flow = None
        if out_f32 is None:
            flow = torch.empty((B, 2, K, 2), dtype=torch.float64, device=self.device)
        elif out_f32.dtype != torch.float32 or not out_f32.is_contiguous() or tuple(out_f32.shape) != (B, 2, K, 2):
            raise ValueError("gen_marker_flow_fused: out_f32 must be a contiguous float32 (B, 2, num_markers, 2) tensor")
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_marker_flow(
                _lib.ptr(x), _lib.ptr(self._surf_ids64), _lib.ptr(self.cam_pos_w), _lib.ptr(self.cam_rot_inv), _lib.ptr(tri), _lib.ptr(wgt),
                self.fx, self.fy, self.cx, self.cy, _lib.ptr(init_uv), _lib.ptr(sel), float(self.tactile_img_width / 2) if self.normalize else 0.0,
                _lib.ptr(curr_uv), _lib.ptr(flow) if flow is not None else None, _lib.ptr(out_f32) if out_f32 is not None else None,
                B, V, M, K, _lib.current_stream_handle(self.device))
        _lib.check(rc, "tacex_fem_marker_flow")
        self.curr_marker_uv = curr_uv
        return flow if out_f32 is None else out_f32

    def gen_marker_flow(self) -> torch.Tensor:
        """(B, 2, num_markers, 2) float64: [initial | current] marker (u, v) pixels (VT:354-413), all envs at once."""
        fused = self.gen_marker_flow_fused()
        if fused is not None:
            return fused
        tri, wgt = self._setup()
        curr_uv = self._project(self.get_surface_vertices_camera(), tri, wgt)
        if self._static:
            return self._gen_marker_flow_static(tri, wgt, curr_uv)
        init_uv = self._project(self.reference_surface_vertices_camera, tri, wgt)
        # VT:382-387 (sic: u is compared with the image HEIGHT and v with the WIDTH); env 0 decides, like the
        # single-env reference; the reference surface is identical in every env
        u0, v0 = init_uv[0, :, 0], init_uv[0, :, 1]
        mask = (u0 > 5) & (u0 < self.tactile_img_height) & (v0 > 5) & (v0 < self.tactile_img_width)
        flow = torch.stack([init_uv, curr_uv], dim=1)[:, :, mask]  # (B,2,M',2)
        if self.marker_lose_tracking_probability > 0.0:
            keep = torch.from_numpy(self._rng.rand(flow.shape[2]) > self.marker_lose_tracking_probability).to(self.device)
            flow = flow[:, :, keep]
        if self.marker_random_noise > 0.0:
            flow = flow + torch.from_numpy(self._rng.randn(*flow.shape[1:]) * self.marker_random_noise).to(self.device)
        n = flow.shape[2]
        if n >= self.num_markers:
            chosen = torch.from_numpy(self._rng.choice(n, self.num_markers, replace=False)).to(self.device)
            ret = flow[:, :, chosen]
        else:  # pad by repeating the last marker (VT:400-405)
            ret = torch.zeros((flow.shape[0], 2, self.num_markers, 2), dtype=flow.dtype, device=self.device)
            ret[:, :, :n] = flow
            if n > 0:
                ret[:, :, n:] = flow[:, :, n - 1:n]
        if self.normalize:
            ret = ret / (self.tactile_img_width / 2) - 1.0
        self.curr_marker_uv = curr_uv
        return ret
