"""CPU oracle for the Taxim optical path (TEST INFRASTRUCTURE - never imported by the product path).

A NumPy/SciPy restatement of the reference algorithm in
`source/tacex/tacex/simulation_approaches/gpu_taxim/sim/taxim_torch.py` (short name TT) and
`.../taxim_impl.py` (TI), `.../taxim_sim.py` (TS).  Each function cites the lines it follows.

Parity status: PINNED.  `tests/golden/*.npz` were produced by importing the reference itself in the
build container (tests/golden/make_golden.py reads /root/reference at run time) and
tests/test_oracle_golden.py checks this file against them with the protocol of SURVEY.md 8(c).

Two blur back-ends:
  * "direct"  - float64 separable correlation with mirror (= torch "reflect") borders. Deterministic:
                a flat region stays exactly flat, so grad_dir == 0 there (idx_dir = 62, idx_mag = 0).
                This is the mode the HIP kernels are compared with.
  * "fft32"   - float32 reflect-pad + 2-D FFT cross-correlation, the way TT:19-44 / TT:381-412 do it.
                Reproduces the reference's roundoff character in flat regions (arbitrary dir bins).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from pathlib import Path

import numpy as np
from scipy import ndimage

F32 = np.float32


# --------------------------------------------------------------------------------------------------
# parameters (TI:17-63, params.json)
# --------------------------------------------------------------------------------------------------
@dataclass
class OracleParams:
    sim: dict
    sensor: dict

    @classmethod
    def load(cls, calib_dir: Path) -> "OracleParams":
        with open(Path(calib_dir) / "params.json") as f:
            p = json.load(f)
        return cls(sim=p["simulator"], sensor=p["sensor"])

    def rel(self, name: str, shape: tuple[int, int]):
        """`<name>_rel` scaled by the image size (TI:33-47): returns (w_val, h_val); tuples element-wise."""
        v = self.sim[name + "_rel"]
        w_val, h_val = v[0], v[1]
        H, W = shape
        w_val = tuple(e * W for e in w_val) if isinstance(w_val, (list, tuple)) else w_val * W
        h_val = tuple(e * H for e in h_val) if isinstance(h_val, (list, tuple)) else h_val * H
        return w_val, h_val

    @property
    def pixmm(self) -> float:
        return self.sensor["pixmm"]

    @property
    def num_bins(self) -> int:
        return self.sensor["num_bins"]

    @property
    def calib_w(self) -> int:
        return self.sensor["w"]

    @property
    def calib_h(self) -> int:
        return self.sensor["h"]


# --------------------------------------------------------------------------------------------------
# Gaussian kernels (TT:362-403)
# --------------------------------------------------------------------------------------------------
def gaussian_kernel_size(sigma: float) -> int:
    """k = odd(round(sqrt(-2 ln(1e-5 sqrt(2 pi) sigma)) * sigma)) with NumPy (banker's) rounding, TT:396-403."""
    eps = 1e-5
    s = np.float64(sigma)
    return int(np.round(np.sqrt(-2 * np.log(eps * np.sqrt(2 * np.pi) * s)) * s).astype(np.int_) // 2 * 2 + 1)


def gaussian_kernel1d(sigma: float, k: int) -> np.ndarray:
    """Normalised float32 taps on linspace(-(k-1)/2, (k-1)/2, k), TT:362-366 (float32 arithmetic)."""
    x = np.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k).astype(F32)
    pdf = np.exp(F32(-0.5) * (x / F32(sigma)) ** 2, dtype=F32)
    return (pdf / pdf.sum(dtype=F32)).astype(F32)


def blur_direct(img: np.ndarray, sigma_wh, ksize_wh=None) -> np.ndarray:
    """Separable correlation over the last two axes in float64, mirror borders (== torch 'reflect', TT:411)."""
    sw, sh = sigma_wh
    kw = gaussian_kernel_size(sw) if ksize_wh is None else ksize_wh[0]
    kh = gaussian_kernel_size(sh) if ksize_wh is None else ksize_wh[1]
    gw = gaussian_kernel1d(sw, kw).astype(np.float64)
    gh = gaussian_kernel1d(sh, kh).astype(np.float64)
    out = np.asarray(img, dtype=np.float64)
    if kh > 1:
        out = ndimage.correlate1d(out, gh, axis=-2, mode="mirror")
    else:
        out = out * gh[0]
    if kw > 1:
        out = ndimage.correlate1d(out, gw, axis=-1, mode="mirror")
    else:
        out = out * gw[0]
    return out


def blur_fft32(img: np.ndarray, sigma_wh, ksize_wh=None) -> np.ndarray:
    """float32 reflect-pad + FFT cross-correlation, keep the valid region (TT:19-44, TT:404-412)."""
    sw, sh = sigma_wh
    kw = gaussian_kernel_size(sw) if ksize_wh is None else ksize_wh[0]
    kh = gaussian_kernel_size(sh) if ksize_wh is None else ksize_wh[1]
    gw = gaussian_kernel1d(sw, kw)
    gh = gaussian_kernel1d(sh, kh)
    k2d = (gh[:, None] * gw[None, :]).astype(F32)
    pw, ph = (kw - 1) // 2, (kh - 1) // 2
    x = np.asarray(img, dtype=F32)
    pad = [(0, 0)] * (x.ndim - 2) + [(ph, ph), (pw, pw)]
    xp = np.pad(x, pad, mode="reflect")
    Hp, Wp = xp.shape[-2:]
    kp = np.zeros((Hp, Wp), dtype=F32)
    kp[:kh, :kw] = k2d
    xf = np.fft.fft2(xp)
    kf = np.fft.fft2(kp)
    out = np.real(np.fft.ifft2(xf * np.conj(kf)))
    return out[..., : Hp - (kh - 1), : Wp - (kw - 1)].astype(F32)


# --------------------------------------------------------------------------------------------------
# bilinear antialiased resize (torchvision.transforms.functional.resize, antialias=True; TT:136-137,159-164)
# --------------------------------------------------------------------------------------------------
def _aa_weights(n_in: int, n_out: int):
    scale = n_in / n_out
    support = scale if scale >= 1.0 else 1.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    rows = []
    for i in range(n_out):
        center = scale * (i + 0.5)
        xmin = max(0, int(center - support + 0.5))
        xmax = min(n_in, int(center + support + 0.5))
        js = np.arange(xmin, xmax)
        w = np.maximum(0.0, 1.0 - np.abs((js - center + 0.5) * inv))
        w = w / w.sum()
        rows.append((xmin, w))
    return rows


def resize_bilinear_aa(img: np.ndarray, out_hw: tuple[int, int]) -> np.ndarray:
    """Separable triangle-filter resize over the last two axes (float64 arithmetic)."""
    x = np.asarray(img, dtype=np.float64)
    H, W = x.shape[-2:]
    oh, ow = out_hw
    if (H, W) == (oh, ow):
        return x.copy()
    wr = _aa_weights(W, ow)
    tmp = np.empty(x.shape[:-1] + (ow,), dtype=np.float64)
    for i, (x0, w) in enumerate(wr):
        tmp[..., i] = (x[..., x0 : x0 + len(w)] * w).sum(-1)
    hr = _aa_weights(H, oh)
    out = np.empty(x.shape[:-2] + (oh, ow), dtype=np.float64)
    for i, (y0, w) in enumerate(hr):
        out[..., i, :] = (tmp[..., y0 : y0 + len(w), :] * w[:, None]).sum(-2)
    return out


# --------------------------------------------------------------------------------------------------
# the oracle proper
# --------------------------------------------------------------------------------------------------
class TaximOracle:
    """Restates TaximTorch.__init__ (TT:50-130) and the per-frame path (TT:174-258, 432-503)."""

    def __init__(self, calib_dir: Path, shape_hw=(240, 320), blur: str = "direct"):
        calib_dir = Path(calib_dir)
        self.p = OracleParams.load(calib_dir)
        self.H, self.W = shape_hw
        self.blur_mode = blur
        self._blur = blur_direct if blur == "direct" else blur_fft32
        ch, cw = self.p.calib_h, self.p.calib_w

        # polynomial table, B<->R swapped on purpose (TT:73-80)
        d = np.load(calib_dir / "polycalib.npz")
        self.poly = (np.stack([d["grad_b"], d["grad_g"], d["grad_r"]], 0) / 255).astype(F32)

        # gel map: blur at calibration resolution, mm, shifted so max == 0, then resized (TT:82-90, 159-164)
        gm = np.load(calib_dir / "gelmap.npy").astype(F32)[None]
        gel = blur_direct(gm, self.p.rel("deform_final_sigma", (ch, cw)))[0] * self.p.pixmm
        self.gel_map_shift = float(gel.max())
        gel_full = (gel - self.gel_map_shift).astype(F32)
        self.gel = resize_bilinear_aa(gel_full, (self.H, self.W)).astype(F32)

        # background (TT:92-94, 414-430, 136-137): f0 BGR 0..255 -> RGB 0..1 CHW
        f0 = np.load(calib_dir / "dataPack.npz", allow_pickle=True)["f0"] / 255
        f0 = np.ascontiguousarray(np.moveaxis(f0.astype(F32), -1, 0)[::-1])
        f0b = blur_direct(f0, self.p.rel("initial_frame_sigma", (ch, cw)))
        d_i = (f0b - f0).mean(0)
        fmp = self.p.sim["frame_mixing_percentage"]
        bg_proc = np.where((d_i < self.p.sim["diff_threshold"])[None], fmp * f0b + (1 - fmp) * f0, f0)
        self.bg = resize_bilinear_aa(bg_proc.astype(F32), (self.H, self.W)).astype(F32)

        # polynomial features in calibration pixel units (TT:139-157)
        ys = (np.arange(self.H, dtype=np.float64) * (ch / self.H)).astype(F32)
        xs = (np.arange(self.W, dtype=np.float64) * (cw / self.W)).astype(F32)
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        self.feat = np.stack([xx * xx, yy * yy, xx * yy, xx, yy, np.ones_like(xx)], -1).astype(F32)

        self.pyr_sigmas = list(zip(*self.p.rel("deform_pyramid_sigma", (self.H, self.W))))  # [(sw, sh)]
        self.final_sigma = self.p.rel("deform_final_sigma", (self.H, self.W))

    # -- TS:115-131 --------------------------------------------------------------------------------
    @staticmethod
    def indentation_depth(hm_mm: np.ndarray, gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024):
        hm_m = np.asarray(hm_mm, dtype=F32) / F32(1000)
        d = hm_m.min(axis=(-2, -1)) - F32(gelpad_to_camera_min_distance)
        d = np.where(d < 0, F32(0), d).astype(F32)
        return np.where(d <= F32(gelpad_height), (F32(gelpad_height) - d) * F32(1000), F32(0)).astype(F32)

    # -- TT:432-441 --------------------------------------------------------------------------------
    @staticmethod
    def shifted_height_map(hm: np.ndarray, press: np.ndarray) -> np.ndarray:
        hm = np.asarray(hm, dtype=F32)
        return (hm - hm.min(axis=(-2, -1), keepdims=True) - np.asarray(press, F32).reshape(-1, 1, 1)).astype(F32)

    # -- TT:443-473 --------------------------------------------------------------------------------
    def gel_pad_deformation(self, S: np.ndarray, return_levels: bool = False):
        S = np.asarray(S, dtype=F32)
        P = -S.min(axis=(-2, -1))
        C = S < 0
        J = np.minimum(S, self.gel[None])
        M = ((J - self.gel[None]) < (-P[:, None, None] * F32(self.p.sim["contact_scale"]))) & C
        wdt = np.float64 if self.blur_mode == "direct" else F32
        Z = J.astype(wdt)
        levels = []
        for sg in self.pyr_sigmas:
            Z = self._blur(Z, sg)
            Z = np.where(M, J.astype(wdt), Z)
            if return_levels:
                levels.append(Z.copy())
        Z = self._blur(Z, self.final_sigma)
        if return_levels:
            return Z, M, J, levels
        return Z, M

    # -- TT:475-503 --------------------------------------------------------------------------------
    def normals(self, z_px: np.ndarray):
        z = np.asarray(z_px)
        H, W = z.shape[-2:]
        dzdx = (z[..., 2:, 1:-1] - z[..., :-2, 1:-1]) / 2.0 * H / self.p.calib_h
        dzdy = (z[..., 1:-1, 2:] - z[..., 1:-1, :-2]) / 2.0 * W / self.p.calib_w
        t = np.sqrt(dzdx**2 + dzdy**2)
        mag = np.arctan(t)
        with np.errstate(invalid="ignore", divide="ignore"):
            dr = np.where(t != 0, np.arctan2(dzdx / t, dzdy / t), 0.0)
        padw = [(0, 0)] * (z.ndim - 2) + [(1, 1), (1, 1)]
        return np.pad(mag, padw, mode="edge"), np.pad(dr, padw, mode="edge")

    def bins(self, mag, dr):
        nb = self.p.num_bins
        x_binr = 0.5 * math.pi / (nb - 1)
        y_binr = 2 * math.pi / (nb - 1)
        dt = mag.dtype
        im = np.floor(mag / dt.type(x_binr)).astype(np.int64)
        idd = np.floor((dr + dt.type(math.pi)) / dt.type(y_binr)).astype(np.int64)
        return im, idd

    # -- TT:225-258 (no-shadow branch) ---------------------------------------------------------------
    def shade(self, Z: np.ndarray, return_all: bool = False):
        wdt = np.float64 if self.blur_mode == "direct" else F32
        z_px = -(np.asarray(Z, wdt) / wdt(self.p.pixmm))
        mag, dr = self.normals(z_px)
        im, idd = self.bins(mag, dr)
        coef = self.poly[:, im, idd]  # (3,B,H,W,6)
        I = (coef.astype(wdt) * self.feat[None, None].astype(wdt)).sum(-1)  # (3,B,H,W)
        rgb = np.clip(np.moveaxis(I, 0, 1) + self.bg[None].astype(wdt), 0, 1)  # (B,3,H,W)
        rgb = np.moveaxis(rgb, 1, -1).astype(F32)  # NHWC like TS:109-111
        if return_all:
            return rgb, mag, dr, im, idd
        return rgb

    # -- TI:153-163 + TT:174-195 ----------------------------------------------------------------------
    def render_direct(self, hm: np.ndarray, press: np.ndarray) -> np.ndarray:
        """(B,H,W) mm height map + (B,) press depth -> (B,H,W,3) float32 RGB in [0,1] (no-shadow path)."""
        S = self.shifted_height_map(hm, press)
        Z, _ = self.gel_pad_deformation(S)
        return self.shade(Z)
