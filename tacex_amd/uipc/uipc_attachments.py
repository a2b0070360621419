"""Soft position constraints that tie gelpad vertices to a rigid body pose - counterpart of
source/tacex_uipc/tacex_uipc/sim/uipc_attachments.py:33-66,364-428 (`aim = R(q) offset + p`, `is_constrained`)."""
from __future__ import annotations

import numpy as np
import torch

from ..utils.configclass import configclass


@configclass
class UipcIsaacAttachmentsCfg:
    constraint_strength_ratio: float = 100.0
    """Stiffness of the constraint relative to the vertex mass (uipc_attachments.py:35-38)."""
    debug_vis: bool = False
    body_name: str = None
    compute_attachment_data: bool = True
    attachment_points_radius: float = 5e-4


def quat_rotate(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Rotate v (B,N,3) by unit quaternions q (B,4) in (w,x,y,z) order."""
    w, xyz = q[:, None, :1], q[:, None, 1:]
    t = 2.0 * torch.cross(xyz.expand_as(v), v, dim=-1)
    return v + w * t + torch.cross(xyz.expand_as(v), t, dim=-1)


class UipcIsaacAttachments:
    """`attachment_points_idx` + `attachment_offsets` (local frame of the rigid body) -> per-step aim positions."""

    def __init__(self, cfg: UipcIsaacAttachmentsCfg, uipc_object, attachment_points_idx: np.ndarray,
                 attachment_offsets: np.ndarray):
        self.cfg = cfg
        self.uipc_object = uipc_object
        self.attachment_points_idx = np.asarray(attachment_points_idx, dtype=np.int64)
        self.attachment_offsets = np.asarray(attachment_offsets, dtype=np.float64)
        if self.attachment_offsets.shape != (len(self.attachment_points_idx), 3):
            raise ValueError("attachment_offsets must be (num_attachment_points, 3)")

    def compute_aim_positions(self, body_pos: torch.Tensor, body_quat: torch.Tensor) -> torch.Tensor:
        """uipc_attachments.py:387-428: aim = T_body * offsets.  body_pos (B,3), body_quat (B,4 wxyz) -> (B,A,3)."""
        off = torch.as_tensor(self.attachment_offsets, device=body_pos.device, dtype=torch.float64)
        off = off[None].expand(body_pos.shape[0], -1, -1)
        return quat_rotate(body_quat.double(), off) + body_pos.double()[:, None, :]
