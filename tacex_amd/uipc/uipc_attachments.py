"""Soft position constraints that tie gelpad vertices to a rigid body pose - counterpart of
source/tacex_uipc/tacex_uipc/sim/uipc_attachments.py:33-66 (cfg), :201-297 (`compute_attachment_data`), :364-385 (the animator
callback that writes `is_constrained` / `aim_position`) and :387-428 (`_compute_aim_positions`: aim = R(q) offset + p).

The reference finds the attached vertices with a PhysX sphere sweep against the rigid body's collider (UA:262-279), keeps
their offsets in the body frame, and every physics step reads the body pose from PhysX, transforms the offsets on the GPU
(float32, `transform_points`), copies the result to the host and lets libuipc's animator write it into the scene.  Here the
collider query is an analytic signed-distance test (box / sphere / callable - there is no PhysX), the per-step part is ONE
HIP launch for all envs (`tacex_fem_set_attachment_targets`) that writes straight into the constraint arrays the Newton
kernels read, and nothing visits the host.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..utils.configclass import configclass


@configclass
class UipcIsaacAttachmentsCfg:
    constraint_strength_ratio: float = 100.0
    """Stiffness of the constraint relative to the vertex mass (uipc_attachments.py:35-38)."""
    debug_vis: bool = False
    body_name: str = None
    compute_attachment_data: bool = True
    attachment_points_radius: float = 5e-4
    """A tet vertex is attached when the rigid collider is within this distance of it (uipc_attachments.py:60-66)."""


def quat_apply_inverse(q: np.ndarray, v: np.ndarray) -> np.ndarray:
    """v rotated by the inverse of the unit quaternion q (w,x,y,z), float32 like UA:289-292."""
    q = np.asarray(q, np.float32)
    v = np.asarray(v, np.float32)
    xyz = q[1:]
    t = np.cross(xyz, v).astype(np.float32) * np.float32(2.0)
    return (v - q[0] * t + np.cross(xyz, t)).astype(np.float32)


class UipcIsaacAttachments:
    """`attachment_points_idx` + `attachment_offsets` (body frame) -> per-step aim positions of the attached vertices."""

    def __init__(self, cfg: UipcIsaacAttachmentsCfg, uipc_object, attachment_points_idx=None, attachment_offsets=None,
                 rigid_collider=None, rigid_pos=(0.0, 0.0, 0.0), rigid_quat=(1.0, 0.0, 0.0, 0.0)):
        self.cfg = cfg
        self.uipc_object = uipc_object
        if attachment_points_idx is None:
            if rigid_collider is None:
                raise ValueError("either precomputed attachment data or a rigid_collider is required (UA:104-131)")
            attachment_offsets, attachment_points_idx, _ = self.compute_attachment_data(
                rigid_collider, uipc_object.points, rigid_pos, rigid_quat, sphere_radius=cfg.attachment_points_radius)
        self.attachment_points_idx = np.asarray(attachment_points_idx, dtype=np.int64)
        self.attachment_offsets = np.asarray(attachment_offsets, dtype=np.float32).reshape(-1, 3)  # float32: UA:288-292
        if self.attachment_offsets.shape != (len(self.attachment_points_idx), 3):
            raise ValueError("attachment_offsets must be (num_attachment_points, 3)")
        self.num_attachment_points_per_obj = len(self.attachment_points_idx)
        self.aim_positions = None  # (B, A, 3) float64 device tensor after the first _compute_aim_positions
        self._dev = None

    # -- UA:201-297 -------------------------------------------------------------------------------------------------
    @staticmethod
    def compute_attachment_data(rigid_collider, tet_points, rigid_pos=(0.0, 0.0, 0.0), rigid_quat=(1.0, 0.0, 0.0, 0.0),
                                sphere_radius: float = 5e-4, max_dist: float = 1e-5):
        """Vertices whose `sphere_radius` ball (swept `max_dist`, UA:263-266) touches the rigid collider, and their offsets in
        the body frame (UA:283-292).  `rigid_collider`: ("box", half_extents) | ("sphere", radius) | callable sdf(local_xyz)
        giving the signed distance [m] to the collider surface in the BODY frame.  Returns (offsets (A,3) f32, idx, positions)."""
        pts = np.asarray(tet_points, np.float64)
        pos = np.asarray(rigid_pos, np.float64)
        q = np.asarray(rigid_quat, np.float32)
        local = np.stack([quat_apply_inverse(q, (p - pos).astype(np.float32)) for p in pts]).astype(np.float64)
        if callable(rigid_collider):
            sd = np.asarray(rigid_collider(local), np.float64)
        elif rigid_collider[0] == "box":
            h = np.asarray(rigid_collider[1], np.float64)
            d = np.abs(local) - h
            sd = np.linalg.norm(np.maximum(d, 0.0), axis=1) + np.minimum(d.max(axis=1), 0.0)
        elif rigid_collider[0] == "sphere":
            sd = np.linalg.norm(local, axis=1) - float(rigid_collider[1])
        else:
            raise ValueError(f"unknown collider {rigid_collider!r}")
        idx = np.where(sd <= sphere_radius + max_dist)[0]
        offsets = np.stack([quat_apply_inverse(q, (pts[i] - pos).astype(np.float32)) for i in idx]).reshape(-1, 3)
        return offsets.astype(np.float32), idx.tolist(), pts[idx]

    # -- UA:387-428 + UA:365-385 ---------------------------------------------------------------------------------------
    def _tables(self, device):
        if self._dev != device:
            self._off_dev = torch.from_numpy(self.attachment_offsets).to(device).contiguous()
            self._idx_dev = torch.from_numpy(self.attachment_points_idx.astype(np.int32)).to(device).contiguous()
            self._dev = device

    def apply(self, uipc_sim, body_pos: torch.Tensor, body_quat: torch.Tensor) -> torch.Tensor:
        """One physics-step callback for all envs: `_compute_aim_positions` (UA:387-428) followed by what the animator does
        with the result (UA:380-385).  body_pos (B,3), body_quat (B,4 wxyz) on the sim's device.  Returns the (B,A,3) aims."""
        dev = uipc_sim.device
        self._tables(dev)
        B, A = uipc_sim.num_envs, self.num_attachment_points_per_obj
        pos = body_pos.to(dev, torch.float32).reshape(B, 3).contiguous()
        quat = body_quat.to(dev, torch.float32).reshape(B, 4).contiguous()
        if self.aim_positions is None or self.aim_positions.shape[0] != B or self.aim_positions.device != dev:
            self.aim_positions = torch.empty((B, A, 3), dtype=torch.float64, device=dev)
        lib = _lib.load_library()
        with torch.cuda.device(dev):
            rc = lib.tacex_fem_set_attachment_targets(
                _lib.ptr(pos), _lib.ptr(quat), _lib.ptr(self._off_dev), _lib.ptr(self._idx_dev), _lib.ptr(uipc_sim.aim_position),
                _lib.ptr(uipc_sim.is_constrained), _lib.ptr(self.aim_positions), B, A, int(uipc_sim.aim_position.shape[1]),
                _lib.current_stream_handle(dev))
        _lib.check(rc, "tacex_fem_set_attachment_targets")
        if getattr(self, "_marked_sim", None) is not uipc_sim:  # the constraint SET changed: the sim rebuilds its coarse operator once
            uipc_sim._mark_constrained(np.asarray(self.attachment_points_idx, dtype=np.int64).reshape(-1))
            self._marked_sim = uipc_sim
        return self.aim_positions

    def _compute_aim_positions(self, uipc_sim, body_pos, body_quat):
        """Reference name (UA:387); same as `apply`."""
        return self.apply(uipc_sim, body_pos, body_quat)
