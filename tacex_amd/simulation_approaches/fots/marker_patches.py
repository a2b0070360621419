"""Marker patch table of the FOTS marker image (fots_marker_sim.py:387-446 `generate_patch_array`).

The reference pre-draws, with OpenCV, one anti-aliased 12 x 12 dot per (sub-pixel phase u, phase v, dot size) -
`patch_array[u, v, w]`, shape (10, 10, 50, 12, 12) uint8 - and `draw_markers` only copies patches around.  The table is DATA:
the HIP kernel (`tacex_fots_marker_image`) takes it as an input, bit for bit.

* `load_patch_array(path)` / `save_patch_array(path, d)`: .npz round trip, so that a table exported once from a TacEx / OpenCV
  installation (`np.savez(path, **generate_patch_array())` there) is reproduced exactly here.
* `generate_patch_array()`: NumPy stand-in for machines without OpenCV (this image has none): same geometry and processing
  chain (filled anti-aliased disk at 10x super-resolution -> 17 x 17 Gaussian, sigma 15 -> cubic down-sample by 10), but
  OpenCV's fixed-point rasteriser and resampler are not reproduced bit for bit - dots differ by a few grey levels on their rim.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

KEYS = ("base_circle_radius", "circle_radius", "size_slot_num", "patch_array", "super_resolution_ratio")


def save_patch_array(path, d: dict) -> None:
    np.savez_compressed(path, **{k: np.asarray(d[k]) for k in KEYS})


def load_patch_array(path) -> dict:
    z = np.load(Path(path))
    d = {
        "base_circle_radius": float(z["base_circle_radius"]),
        "circle_radius": int(z["circle_radius"]),
        "size_slot_num": int(z["size_slot_num"]),
        "patch_array": np.ascontiguousarray(z["patch_array"], dtype=np.uint8),
        "super_resolution_ratio": int(z["super_resolution_ratio"]),
    }
    check_patch_array(d)
    return d


def check_patch_array(d: dict) -> None:
    pa = d["patch_array"]
    sr, s = int(d["super_resolution_ratio"]), int(d["size_slot_num"])
    if pa.dtype != np.uint8 or pa.shape != (sr, sr, s, 12, 12):
        raise ValueError(f"patch_array must be uint8 ({sr}, {sr}, {s}, 12, 12), got {pa.dtype} {pa.shape}")


def _cubic_weights(t: np.ndarray, a: float = -0.75) -> np.ndarray:
    """OpenCV's bicubic kernel (a = -0.75) for the 4 taps around a sample at fractional offset t."""
    x = np.stack([1 + t, t, 1 - t, 2 - t], -1)
    w = np.where(x <= 1, (a + 2) * x**3 - (a + 3) * x**2 + 1, a * x**3 - 5 * a * x**2 + 8 * a * x - 4 * a)
    return w


def generate_patch_array(super_resolution_ratio: int = 10, _phases=None) -> dict:
    """`_phases`: optional list of (u, v) sub-pixel phases to draw (tests); the others stay zero."""
    from scipy.ndimage import convolve1d

    circle_radius, size_slot_num, base_circle_radius = 3, 50, 1.5
    sr = super_resolution_ratio
    n = 4 * circle_radius * sr  # 120
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    k = np.arange(17) - 8.0
    g = np.exp(-0.5 * (k / 15.0) ** 2)
    g /= g.sum()
    # cubic down-sample by `sr`: destination pixel d samples source coordinate (d + 0.5) * sr - 0.5
    src = (np.arange(4 * circle_radius) + 0.5) * sr - 0.5
    i0 = np.floor(src).astype(int)
    cw = _cubic_weights(src - i0)
    idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n - 1)
    radii = np.array([round(base_circle_radius * sr + w) for w in range(size_slot_num)], np.float64)
    patch = np.zeros((sr, sr, size_slot_num, 4 * circle_radius, 4 * circle_radius), np.uint8)
    for u in range(sr):
        for v in range(sr):
            if _phases is not None and (u, v) not in _phases:
                continue
            cx, cy = 2 * circle_radius * sr + u, 2 * circle_radius * sr + v
            dist = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
            cov = np.clip(radii[:, None, None] + 0.5 - dist[None], 0.0, 1.0)   # anti-aliased filled disk, one per size slot
            img = np.rint(255.0 * (1.0 - cov))
            img = convolve1d(convolve1d(img, g, axis=1, mode="mirror"), g, axis=2, mode="mirror")  # 17 x 17 Gaussian, reflect-101
            img = np.rint(img)
            rows = (img[:, idx, :] * cw[None, :, :, None]).sum(2)             # (S, 12, n)
            low = (rows[:, :, idx] * cw[None, None, :, :]).sum(3)             # (S, 12, 12)
            patch[u, v] = np.clip(np.rint(low), 0, 255).astype(np.uint8)
    return {"base_circle_radius": base_circle_radius, "circle_radius": circle_radius, "size_slot_num": size_slot_num,
            "patch_array": patch, "super_resolution_ratio": sr}
