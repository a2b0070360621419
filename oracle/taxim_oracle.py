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

        # shadow calibration (TT:96-126): directions, fan of 4 rays around each, table padded with +inf to 51 steps,
        # BGR -> RGB flip; the "extra empty entry" concat of the reference does not change the shape (24 heights)
        sd = np.load(calib_dir / "shadowTable.npz", allow_pickle=True)
        direction = sd["shadowDirections"].astype(F32)
        fan_angle = self.p.sim["fan_angle"]
        n_fan = int(fan_angle * 2 / self.p.sim["fan_precision"])
        self.fan = (direction[:, None] + np.linspace(-fan_angle, fan_angle, n_fan).astype(F32)[None, :]).astype(F32)
        tab = sd["shadowTable"][::-1]  # flip channel axis
        maxlen = max(len(e) for e in tab.reshape(-1))
        self.shadow_table = (np.array([list(e) + [np.inf] * (maxlen - len(e)) for e in tab.reshape(-1)], dtype=F32)
                             .reshape(tab.shape + (maxlen,)) / F32(255))
        self.shadow_depth_0 = 0.4

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

    # -- TT:260-346 (shadow branch) ---------------------------------------------------------------------
    @staticmethod
    def _box_dilate_same(mask_f: np.ndarray, kh: int, kw: int) -> np.ndarray:
        """conv2d(ones((kh,kw)), padding='same'): torch pads (k-1)//2 before and the rest after (TT:268-272)."""
        H, W = mask_f.shape[-2:]
        top, left = (kh - 1) // 2, (kw - 1) // 2
        pad = [(0, 0)] * (mask_f.ndim - 2) + [(top, kh - 1 - top), (left, kw - 1 - left)]
        mp = np.pad(mask_f, pad)
        out = np.zeros_like(mask_f)
        for i in range(kh):
            for j in range(kw):
                out += mp[..., i : i + H, j : j + W]
        return out

    def shadow_attachment_rounds(self):
        ks = np.array(self.p.rel("shadow_attachment_kernel_size", (self.H, self.W)))  # (w, h)
        total = np.round(ks * 2).astype(np.int_)
        first = total // 2
        return [np.maximum(1, first), np.maximum(1, total - first)]  # each (kw, kh)

    def shadow_map(self, Z: np.ndarray, M: np.ndarray):
        """(B,H,W,3) per-pixel / channel minimum of the shadow-table samples (+inf: no sample) = `shadow_img` of TT:324-336,
        and the gradient direction map it was marched with."""
        return self.shade_with_shadow(Z, M, _return_shadow_map=True)

    def shade_with_shadow(self, Z: np.ndarray, M: np.ndarray, _return_shadow_map: bool = False) -> np.ndarray:
        """Deformed gel (B,H,W) + shrunken contact mask -> (B,H,W,3) RGB with cast shadows."""
        H, W = self.H, self.W
        B = Z.shape[0]
        Zf = np.asarray(Z, F32)
        z_px = -(Zf / F32(self.p.pixmm))
        mag, dr = self.normals(z_px)
        im, idd = self.bins(mag, dr)
        coef = self.poly[:, im, idd]
        sim = np.moveaxis((coef * self.feat[None, None]).sum(-1, dtype=F32), 0, 1).astype(F32)  # (B,3,H,W)
        dil = M.astype(F32)
        for (kw, kh) in self.shadow_attachment_rounds():
            dil = self._box_dilate_same(dil, int(kh), int(kw))
        boundary = (dil != 0) & ~M
        bi, yi, xi = np.nonzero(boundary)
        norm_idx = np.floor((dr[boundary].astype(F32) + F32(math.pi)) / F32(self.p.sim["discretize_precision"])).astype(np.int64)
        zpx = (Zf / F32(self.p.pixmm)).astype(F32)  # deformed_gel_px
        contact_px = ((self.gel[None] - Zf) / F32(self.p.pixmm)).astype(F32)[boundary]
        hidx = np.floor((contact_px * F32(self.p.pixmm) - F32(self.shadow_depth_0)) / F32(self.p.sim["height_precision"])).astype(np.int64) + 6
        max_h = self.shadow_table.shape[2] - 1
        hidx[(hidx < 0) | (hidx >= max_h)] = max_h
        sel = self.shadow_table[:, norm_idx, hidx]  # (3,N,51)
        thetas = self.fan[norm_idx]                 # (N,4)
        nstep = sel.shape[-1]
        steps = (np.arange(nstep) + 1).astype(F32)
        step_w, step_h = self.p.rel("shadow_step", (H, W))  # (w_val, h_val); x uses [1], y uses [0] (TT:300-305)
        # cos / sin of the float32 fan table (TT:299,303 take them of the gathered angles; taking them of the table first and
        # gathering is the same numbers and lets the device use the very same bits, tacex_amd/calibration.py:build_shadow_tables)
        cos_t, sin_t = np.cos(self.fan).astype(F32)[norm_idx], np.sin(self.fan).astype(F32)[norm_idx]
        sx = (xi[:, None, None].astype(F32) + ((F32(step_h) * steps)[None, None, :] * cos_t[:, :, None]).astype(F32)).astype(F32)
        sy = (yi[:, None, None].astype(F32) + ((F32(step_w) * steps)[None, None, :] * sin_t[:, :, None]).astype(F32)).astype(F32)
        sx = np.trunc(sx).astype(np.int64)
        sy = np.trunc(sy).astype(np.int64)
        cx, cy = np.clip(sx, 0, W - 1), np.clip(sy, 0, H - 1)
        valid = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H) & (zpx[bi, yi, xi][:, None, None] < zpx[bi[:, None, None], cy, cx])
        shadow = np.full((3, B * H * W), np.inf, F32)
        n_i, f_i, s_i = np.nonzero(valid)
        flat = (bi[n_i] * H + sy[valid]) * W + sx[valid]
        for c in range(3):
            np.minimum.at(shadow[c], flat, sel[c, n_i, s_i])
        if _return_shadow_map:
            return np.moveaxis(shadow.reshape(3, B, H, W), 0, -1), dr
        sim = np.minimum(sim, np.moveaxis(shadow.reshape(3, B, H, W), 0, 1))
        wdt = np.float64 if self.blur_mode == "direct" else F32
        s1 = self._blur(sim.astype(wdt), self.p.rel("shadow_blur_sigma", (H, W)))
        s2 = self._blur(s1 + self.bg[None].astype(wdt), self.final_sigma)
        return np.moveaxis(np.clip(s2, 0, 1), 1, -1).astype(F32)

    # -- TI:153-163 + TT:174-195 ----------------------------------------------------------------------
    def render_direct(self, hm: np.ndarray, press: np.ndarray, with_shadow: bool = False) -> np.ndarray:
        """(B,H,W) mm height map + (B,) press depth -> (B,H,W,3) float32 RGB in [0,1]."""
        S = self.shifted_height_map(hm, press)
        Z, M = self.gel_pad_deformation(S)
        return self.shade_with_shadow(Z, M) if with_shadow else self.shade(Z)
