"""Seeded synthetic contact depth maps in the GelSight camera-depth convention.

Depth is in millimetres: background = far clip (29.0 mm), gel top at 28.5 mm (= 24 + 4.5), indenters
reach down towards the camera. Used by tests, the golden-vector generator and bench.py; it produces
inputs only and contains no part of the tactile pipeline.
"""
from __future__ import annotations

import math

import torch

FAR_CLIP_MM = 29.0
GEL_TOP_MM = 28.5  # gelpad_to_camera_min_distance (24 mm) + gelpad_height (4.5 mm)
PIXMM = 0.0295


def synthetic_depth_maps(
    num_frames: int,
    height: int = 240,
    width: int = 320,
    seed: int = 0,
    device: str | torch.device = "cpu",
    flat_fraction: float = 0.1,
    kinds: tuple[str, ...] = ("sphere", "cylinder", "edge", "two_spheres"),
    chunk: int = 256,
) -> tuple[torch.Tensor, torch.Tensor]:
    """Returns (depth_mm (B,H,W) float32 on `device`, indent_mm (B,) float32 = intended press depth).

    Every frame gets a seeded random indenter; `flat_fraction` of the frames have no contact at all
    (pure far-clip background), which is what the reference's benchmark counts separately
    (run_ball_rolling_experiment.py:238-244).
    """
    g = torch.Generator().manual_seed(seed)
    B = num_frames
    u = torch.rand((B, 8), generator=g)
    kind_id = torch.randint(0, len(kinds), (B,), generator=g)
    flat = torch.rand((B,), generator=g) < flat_fraction
    scale = height / 240.0  # indenter sizes follow the image size (pixel pitch scales with resolution)
    pixmm = PIXMM / scale
    indent = (0.2 + 1.3 * u[:, 0]).float()
    indent = torch.where(flat, torch.zeros_like(indent), indent)
    out = torch.empty((B, height, width), dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(
        torch.arange(height, dtype=torch.float32, device=device),
        torch.arange(width, dtype=torch.float32, device=device),
        indexing="ij",
    )
    for s in range(0, B, chunk):
        e = min(B, s + chunk)
        uu = u[s:e].to(device)
        ind = indent[s:e].to(device).view(-1, 1, 1)
        kid = kind_id[s:e].to(device).view(-1, 1, 1)
        fl = flat[s:e].to(device).view(-1, 1, 1)
        r = ((0.15 + 0.20 * uu[:, 1]) * height).view(-1, 1, 1)  # radius in px
        cx = ((0.3 + 0.4 * uu[:, 2]) * width).view(-1, 1, 1)
        cy = ((0.3 + 0.4 * uu[:, 3]) * height).view(-1, 1, 1)
        ang = (math.pi * uu[:, 4]).view(-1, 1, 1)
        dx, dy = xx[None] - cx, yy[None] - cy
        # sphere cap height (mm) above its lowest point
        sph = (r - torch.sqrt(torch.clamp(r * r - dx * dx - dy * dy, min=0.0))) * pixmm
        sph = torch.where(dx * dx + dy * dy <= r * r, sph, torch.full_like(sph, 1e3))
        # cylinder lying in the image plane, axis along `ang`
        dn = -dx * torch.sin(ang) + dy * torch.cos(ang)
        rc = 0.5 * r
        cyl = (rc - torch.sqrt(torch.clamp(rc * rc - dn * dn, min=0.0))) * pixmm
        cyl = torch.where(dn.abs() <= rc, cyl, torch.full_like(cyl, 1e3))
        # wedge edge: a ridge with 45 degree flanks, limited length
        dt = dx * torch.cos(ang) + dy * torch.sin(ang)
        edg = dn.abs() * pixmm
        edg = torch.where((dt.abs() <= r) & (dn.abs() <= 0.6 * r), edg, torch.full_like(edg, 1e3))
        # two spheres
        cx2 = cx + (0.5 + uu[:, 5].view(-1, 1, 1)) * r
        cy2 = cy + (uu[:, 6].view(-1, 1, 1) - 0.5) * r
        r2 = 0.6 * r
        d2x, d2y = xx[None] - cx2, yy[None] - cy2
        sp2 = (r2 - torch.sqrt(torch.clamp(r2 * r2 - d2x * d2x - d2y * d2y, min=0.0))) * pixmm + 0.1
        sp2 = torch.where(d2x * d2x + d2y * d2y <= r2 * r2, sp2, torch.full_like(sp2, 1e3))
        two = torch.minimum(sph, sp2)
        shapes = {"sphere": sph, "cylinder": cyl, "edge": edg, "two_spheres": two}
        prof = torch.full_like(sph, 1e3)
        for i, k in enumerate(kinds):
            prof = torch.where(kid == i, shapes[k], prof)
        depth = GEL_TOP_MM - ind + prof  # lowest indenter point is `indent` below the gel top
        depth = torch.clamp(depth, max=FAR_CLIP_MM)
        depth = torch.where(fl, torch.full_like(depth, FAR_CLIP_MM), depth)
        out[s:e] = depth
    return out, indent.to(device)


def dense_contact_depth_maps(num_frames: int, height: int = 240, width: int = 320, seed: int = 0,
                             device: str | torch.device = "cpu") -> tuple[torch.Tensor, torch.Tensor]:
    """Worst-case input for the data-dependent shortcuts of the pipeline: a wavy plate pressed over the WHOLE sensor, so every
    frame, every row and (nearly) every pixel is in contact - no zero band to skip, no flat wave, the height map is needed on
    every row and every pixel's table record lies beyond magnitude bin 0.  Returns (depth_mm (B,H,W), indent_mm (B,))."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((num_frames, 6), generator=g).to(device)
    indent = (0.6 + 0.9 * u[:, 0]).float()
    yy, xx = torch.meshgrid(torch.arange(height, dtype=torch.float32, device=device),
                            torch.arange(width, dtype=torch.float32, device=device), indexing="ij")
    s = height / 240.0
    out = torch.empty((num_frames, height, width), dtype=torch.float32, device=device)
    for b0 in range(0, num_frames, 256):
        uu = u[b0:b0 + 256]
        fx = (0.04 + 0.05 * uu[:, 1]).view(-1, 1, 1) / s
        fy = (0.05 + 0.05 * uu[:, 2]).view(-1, 1, 1) / s
        wav = 0.15 * (1.0 + torch.sin(fx * xx[None] + 6.28 * uu[:, 3].view(-1, 1, 1)) * torch.cos(fy * yy[None] + 6.28 * uu[:, 4].view(-1, 1, 1)))
        out[b0:b0 + 256] = GEL_TOP_MM - indent[b0:b0 + 256].view(-1, 1, 1) + wav
    return out, indent
