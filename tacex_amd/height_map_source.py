"""Height-map sources: what fills `output["height_map"]` when no camera depth is injected.

The reference renders the gel pad with an IsaacLab `TiledCamera` and converts the depth image
(gelsight_sensor.py:229-263, 581-593).  `IndenterHeightMapSource` replaces that round trip for primitive indenters: one
HIP launch rasterises the contact geometry of every env straight into the height map and leaves the per-frame minimum
and the indentation depth (taxim_sim.py:115-131) behind, exactly what `tacex_height_map_from_depth` would have produced
from the rendered depth.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

KINDS = {"none": -1.0, "sphere": 0.0, "cylinder": 1.0, "edge": 2.0, "two_spheres": 3.0}


class IndenterHeightMapSource:
    """Per-env analytic indenter: `params` is a (num_envs, 8) float32 device tensor
    [kind, cx_px, cy_px, r_px, angle_rad, press_mm, cx2_px, cy2_px] that the caller updates in place between steps."""

    def __init__(self, num_envs: int, device, pixmm: float = 0.0295, gel_top_mm: float = 28.5, far_clip_mm: float = 29.0):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.TacexHipError("IndenterHeightMapSource needs an AMD GPU device (no CPU fallback)")
        self.params = torch.zeros((num_envs, 8), dtype=torch.float32, device=dev)
        self.params[:, 0] = KINDS["none"]
        self.pixmm, self.gel_top_mm, self.far_clip_mm = float(pixmm), float(gel_top_mm), float(far_clip_mm)
        self._lib = _lib.load_library()

    def set(self, kind, cx, cy, r, angle=0.0, press_mm=0.0, cx2=0.0, cy2=0.0, env_ids=slice(None)):
        """Convenience setter; arguments are scalars or per-env tensors, `kind` a name from KINDS or a tensor of codes."""
        k = KINDS[kind] if isinstance(kind, str) else kind
        for col, v in enumerate((k, cx, cy, r, angle, press_mm, cx2, cy2)):
            self.params[env_ids, col] = v if not isinstance(v, torch.Tensor) else v.to(self.params)

    def fill(self, hm: torch.Tensor, frame_min: torch.Tensor, indent: torch.Tensor | None, gelpad_height: float,
             gelpad_to_camera_min_distance: float):
        """hm (B, H, W) mm, frame_min (B,), indent (B,) or None - all written in one launch on the current stream."""
        B, H, W = hm.shape
        with torch.cuda.device(hm.device):
            rc = self._lib.tacex_height_map_from_indenters(
                _lib.ptr(self.params), self.pixmm, self.gel_top_mm, self.far_clip_mm, float(gelpad_height),
                float(gelpad_to_camera_min_distance), _lib.ptr(hm), _lib.ptr(frame_min), _lib.ptr(indent), B, H, W,
                _lib.current_stream_handle(hm.device))
        _lib.check(rc, "tacex_height_map_from_indenters")


class MeshDepthSource:
    """Camera depth of a rigid triangle mesh per env (arbitrary indenters): a callable for
    `cfg.sensor_camera_cfg.depth_source` that stands in for the IsaacLab TiledCamera read-out of the reference
    (gelsight_sensor.py:229-263): `source()` -> (num_envs, H, W) float32 "distance_to_image_plane" depth in metres, inf where
    nothing lies inside the clipping range.  The caller updates `pos` (num_envs, 3) and `quat` (num_envs, 4, wxyz) in place -
    the pose of the object in the CAMERA frame (x right, y down, z along the optical axis)."""

    def __init__(self, verts, tris, num_envs: int, device, resolution=(320, 240), intrinsics=(340.0, 325.0, 160.0, 125.0),
                 clipping_range=(0.024, 0.029)):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.TacexHipError("MeshDepthSource needs an AMD GPU device (no CPU fallback)")
        self.verts = torch.as_tensor(verts, dtype=torch.float32).reshape(-1, 3).contiguous().to(dev)
        self.tris = torch.as_tensor(tris, dtype=torch.int32).reshape(-1, 3).contiguous().to(dev)
        if self.tris.numel() == 0 or int(self.tris.min()) < 0 or int(self.tris.max()) >= self.verts.shape[0]:
            raise ValueError("MeshDepthSource: triangle indices out of range")
        self.W, self.H = int(resolution[0]), int(resolution[1])
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in intrinsics)
        self.near, self.far = float(clipping_range[0]), float(clipping_range[1])
        self.pos = torch.zeros((num_envs, 3), dtype=torch.float32, device=dev)
        self.pos[:, 2] = 1.0  # out of range until the caller places the object
        self.quat = torch.zeros((num_envs, 4), dtype=torch.float32, device=dev)
        self.quat[:, 0] = 1.0
        self.depth = torch.empty((num_envs, self.H, self.W), dtype=torch.float32, device=dev)
        v = self.verts.double().cpu()
        c = 0.5 * (v.min(0).values + v.max(0).values)
        self._bsphere = (C.c_float * 4)(float(c[0]), float(c[1]), float(c[2]), float((v - c).norm(dim=1).max()) * 1.0001)
        self._lib = _lib.load_library()

    def __call__(self) -> torch.Tensor:
        with torch.cuda.device(self.depth.device):
            rc = self._lib.tacex_depth_from_mesh(
                _lib.ptr(self.verts), _lib.ptr(self.tris), int(self.verts.shape[0]), int(self.tris.shape[0]),
                _lib.ptr(self.pos), _lib.ptr(self.quat),
                self.fx, self.fy, self.cx, self.cy, self.near, self.far, C.cast(self._bsphere, C.c_void_p), _lib.ptr(self.depth),
                int(self.depth.shape[0]),
                self.H, self.W, _lib.current_stream_handle(self.depth.device))
        _lib.check(rc, "tacex_depth_from_mesh")
        return self.depth
