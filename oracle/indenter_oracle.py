"""CPU oracle of the analytic-indenter height-map source - TEST INFRASTRUCTURE ONLY.

The reference has no such function: its height map comes from the IsaacLab TiledCamera depth render
(gelsight_sensor.py:229-263 -> _get_height_map, gelsight_sensor.py:581-593), which cannot run here.  This restates, in
NumPy float32, the scene model the HIP kernel `indenter_height_map_kernel` implements (csrc/taxim_kernels.hip) plus the
reference's indentation-depth rule (taxim_sim.py:115-131).  PARITY UNPINNED against the reference (nothing to pin to);
pinned to `tacex_amd.utils.synthetic.synthetic_depth_maps`, the generator the golden vectors were made with.
"""
import numpy as np

F32 = np.float32


def indenter_height_map(desc, H, W, pixmm, gel_top_mm=28.5, far_clip_mm=29.0):
    """desc (B, 8) = [kind, cx, cy, r, angle, press_mm, cx2, cy2] -> (B, H, W) float32 height map in mm."""
    d = np.asarray(desc, F32)
    B = d.shape[0]
    yy, xx = np.meshgrid(np.arange(H, dtype=F32), np.arange(W, dtype=F32), indexing="ij")
    out = np.empty((B, H, W), F32)
    big = F32(1e3)
    for b in range(B):
        kind, cx, cy, r, ang, press, cx2, cy2 = d[b]
        if kind < 0:
            out[b] = F32(far_clip_mm)
            continue
        dx, dy = xx - cx, yy - cy
        sn, cs = F32(np.sin(ang)), F32(np.cos(ang))
        k = int(kind)
        if k in (0, 3):
            q = dx * dx + dy * dy
            prof = np.where(q <= r * r, (r - np.sqrt(np.maximum(r * r - q, F32(0)))) * F32(pixmm), big)
            if k == 3:
                r2 = F32(0.6) * r
                ex, ey = xx - cx2, yy - cy2
                q2 = ex * ex + ey * ey
                p2 = np.where(q2 <= r2 * r2, (r2 - np.sqrt(np.maximum(r2 * r2 - q2, F32(0)))) * F32(pixmm) + F32(0.1), big)
                prof = np.minimum(prof, p2)
        else:
            dn = -dx * sn + dy * cs
            if k == 1:
                rc = F32(0.5) * r
                prof = np.where(np.abs(dn) <= rc, (rc - np.sqrt(np.maximum(rc * rc - dn * dn, F32(0)))) * F32(pixmm), big)
            else:
                dt = dx * cs + dy * sn
                prof = np.where((np.abs(dt) <= r) & (np.abs(dn) <= F32(0.6) * r), np.abs(dn) * F32(pixmm), big)
        out[b] = np.minimum((F32(gel_top_mm) - press) + prof.astype(F32), F32(far_clip_mm))
    return out


def indentation_depth(hm_mm, gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024):
    """taxim_sim.py:115-131 on a (B, H, W) height map in mm."""
    m = hm_mm.reshape(hm_mm.shape[0], -1).min(axis=1).astype(F32)
    d = np.maximum(m / F32(1000.0) - F32(gelpad_to_camera_min_distance), F32(0))
    return m, np.where(d <= F32(gelpad_height), (F32(gelpad_height) - d) * F32(1000.0), F32(0)).astype(F32)
