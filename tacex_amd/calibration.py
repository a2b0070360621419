"""Host-side, one-off preparation of the Taxim calibration tables (init only, not the hot path).

Mirrors what the reference does once in `TaximTorch.__init__` and its lru-cached getters
(gpu_taxim/sim/taxim_torch.py:50-164) and the `_rel` parameter scaling of
gpu_taxim/sim/taxim_impl.py:17-63,183-202.  The per-frame work lives in csrc/*.hip.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any

import numpy as np

F32 = np.float32

CALIB_GELSIGHT_MINI = Path(__file__).resolve().parent / "assets" / "calib" / "gsmini_640x480"
CALIB_GELSIGHT = CALIB_GELSIGHT_MINI  # name used by the reference's sim package (sim/calibration.py)


# -- parameters (taxim_impl.py:17-63) ------------------------------------------------------------------
@dataclass(frozen=True)
class SimulatorParameters:
    initial_frame_sigma_rel: tuple
    frame_mixing_percentage: float
    diff_threshold: int
    contact_scale: float
    deform_pyramid_sigma_rel: tuple
    shadow_blur_sigma_rel: tuple
    deform_final_sigma_rel: tuple
    shadow_step_rel: tuple
    height_precision: float
    discretize_precision: float
    fan_angle: float
    fan_precision: float
    shadow_attachment_kernel_size_rel: tuple

    def __getattr__(self, item):
        # every `<name>_rel` parameter is exposed as `<name>(shape)` scaled by the image size:
        # element 0 by the width, element 1 by the height (taxim_impl.py:33-47)
        rel_name = f"{item}_rel"
        if not item.endswith("_rel") and rel_name in self.__dataclass_fields__:
            value = object.__getattribute__(self, rel_name)

            def scaled(shape: tuple[int, int]):
                assert len(shape) == 2
                w_val, h_val = value[0], value[1]
                w_val = tuple(e * shape[1] for e in w_val) if isinstance(w_val, tuple) else w_val * shape[1]
                h_val = tuple(e * shape[0] for e in h_val) if isinstance(h_val, tuple) else h_val * shape[0]
                return w_val, h_val

            return scaled
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{item}'")


@dataclass
class SensorParameters:
    w: int
    h: int
    pixmm: float
    num_bins: int

    @property
    def width(self) -> int:
        return self.w

    @property
    def height(self) -> int:
        return self.h


def _lists_to_tuples(obj):
    if isinstance(obj, list):
        return tuple(_lists_to_tuples(i) for i in obj)
    if isinstance(obj, dict):
        return {k: _lists_to_tuples(v) for k, v in obj.items()}
    return obj


def _update_dict_recursive(default: dict, update: dict) -> dict:
    """Recursive override; unknown keys raise ValueError like taxim_impl.py:183-202."""
    unknown = [k for k in update if k not in default]
    if unknown:
        raise ValueError(f"Unknown key(s): {', '.join(map(str, unknown))}")
    return {
        k: (_update_dict_recursive(default[k], update[k]) if isinstance(default[k], dict) and k in update
            else update.get(k, default[k]))
        for k in default
    }


def load_params(calib_folder: Path, params: dict[str, dict[str, Any]] | None = None):
    with (Path(calib_folder) / "params.json").open() as f:
        default = json.load(f)
    merged = _update_dict_recursive(default, params) if params is not None else default
    return (SimulatorParameters(**_lists_to_tuples(merged["simulator"])),
            SensorParameters(**_lists_to_tuples(merged["sensor"])))


# -- Gaussian kernels (taxim_torch.py:362-403) -----------------------------------------------------------
def gaussian_kernel_size(sigma: float) -> int:
    eps = 1e-5
    s = np.float64(sigma)
    return int(np.round(np.sqrt(-2 * np.log(eps * np.sqrt(2 * np.pi) * s)) * s).astype(np.int_) // 2 * 2 + 1)


def gaussian_taps(sigma: float, k: int) -> np.ndarray:
    x = np.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k).astype(F32)
    pdf = np.exp(F32(-0.5) * (x / F32(sigma)) ** 2, dtype=F32)
    return np.ascontiguousarray((pdf / pdf.sum(dtype=F32)).astype(F32))


def _correlate_mirror(img: np.ndarray, taps: np.ndarray, axis: int) -> np.ndarray:
    """1-D correlation with torch-'reflect' borders along `axis` (float64), for the init-time blurs."""
    k = len(taps)
    if k == 1:
        return img * taps[0]
    r = (k - 1) // 2
    pad = [(0, 0)] * img.ndim
    pad[axis] = (r, r)
    xp = np.pad(img, pad, mode="reflect")
    out = np.zeros_like(img, dtype=np.float64)
    n = img.shape[axis]
    for t in range(k):
        sl = [slice(None)] * img.ndim
        sl[axis] = slice(t, t + n)
        out += taps[t] * xp[tuple(sl)]
    return out


def gaussian_blur_host(img: np.ndarray, sigma_wh) -> np.ndarray:
    kw, kh = gaussian_kernel_size(sigma_wh[0]), gaussian_kernel_size(sigma_wh[1])
    out = np.asarray(img, np.float64)
    out = _correlate_mirror(out, gaussian_taps(sigma_wh[1], kh).astype(np.float64), img.ndim - 2)
    out = _correlate_mirror(out, gaussian_taps(sigma_wh[0], kw).astype(np.float64), img.ndim - 1)
    return out


# -- bilinear antialiased resize (torchvision resize semantics) ----------------------------------------------
def _aa_rows(n_in: int, n_out: int):
    scale = n_in / n_out
    support = scale if scale >= 1.0 else 1.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    for i in range(n_out):
        center = scale * (i + 0.5)
        lo = max(0, int(center - support + 0.5))
        hi = min(n_in, int(center + support + 0.5))
        js = np.arange(lo, hi)
        w = np.maximum(0.0, 1.0 - np.abs((js - center + 0.5) * inv))
        yield i, lo, w / w.sum()


def resize_bilinear_aa_host(img: np.ndarray, out_hw: tuple[int, int]) -> np.ndarray:
    x = np.asarray(img, np.float64)
    H, W = x.shape[-2:]
    oh, ow = out_hw
    if (H, W) == (oh, ow):
        return x.copy()
    tmp = np.empty(x.shape[:-1] + (ow,), np.float64)
    for i, lo, w in _aa_rows(W, ow):
        tmp[..., i] = (x[..., lo:lo + len(w)] * w).sum(-1)
    out = np.empty(x.shape[:-2] + (oh, ow), np.float64)
    for i, lo, w in _aa_rows(H, oh):
        out[..., i, :] = (tmp[..., lo:lo + len(w), :] * w[:, None]).sum(-2)
    return out


def torch_linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """float32 torch.linspace: ascending from `start` for the first half, descending from `end` after."""
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    idx = np.arange(steps)
    half = steps // 2
    lo = (start + step * idx.astype(F32)).astype(F32)
    hi = (end - step * (steps - idx - 1).astype(F32)).astype(F32)
    return np.where(idx < half, lo, hi).astype(F32)


# -- the table bundle -------------------------------------------------------------------------------------
@dataclass
class TaximTables:
    """Everything the device context needs for one tactile resolution (H, W)."""

    height: int
    width: int
    sim_params: SimulatorParameters
    sensor_params: SensorParameters
    ksize_w: list[int] = field(default_factory=list)
    ksize_h: list[int] = field(default_factory=list)
    taps_w: list[np.ndarray] = field(default_factory=list)
    taps_h: list[np.ndarray] = field(default_factory=list)
    poly: np.ndarray = None         # (3, nb, nb, 6) f32
    gel_map: np.ndarray = None      # (H, W) f32
    gel_map_shift: float = 0.0
    background: np.ndarray = None   # (3, H, W) f32
    background_full: np.ndarray = None  # (3, calib_h, calib_w) f32 (the reference's `background_img`)
    feat_x: np.ndarray = None
    feat_y: np.ndarray = None
    shadow: dict = None             # shadow-branch tables (build_shadow_tables), filled on demand


def build_taxim_tables(calib_folder: Path, shape_hw: tuple[int, int],
                       params: dict[str, dict[str, Any]] | None = None) -> TaximTables:
    calib_folder = Path(calib_folder)
    sim, sensor = load_params(calib_folder, params)
    H, W = shape_hw
    ch, cw = sensor.height, sensor.width
    t = TaximTables(height=H, width=W, sim_params=sim, sensor_params=sensor)

    # pyramid + final blur kernels (taxim_torch.py:464-471)
    pw, ph = sim.deform_pyramid_sigma((H, W))
    fw, fh = sim.deform_final_sigma((H, W))
    for sw, sh in list(zip(pw, ph)) + [(fw, fh)]:
        kw, kh = gaussian_kernel_size(sw), gaussian_kernel_size(sh)
        t.ksize_w.append(kw)
        t.ksize_h.append(kh)
        t.taps_w.append(gaussian_taps(sw, kw))
        t.taps_h.append(gaussian_taps(sh, kh))

    # polynomial table with the intentional b<->r swap (taxim_torch.py:73-80)
    d = np.load(calib_folder / "polycalib.npz")
    t.poly = np.ascontiguousarray((np.stack([d["grad_b"], d["grad_g"], d["grad_r"]], 0) / 255).astype(F32))
    nb = sensor.num_bins
    if t.poly.shape != (3, nb, nb, 6):
        raise ValueError(f"polycalib.npz has shape {t.poly.shape}, expected (3, {nb}, {nb}, 6)")

    # gel map (taxim_torch.py:82-90,159-164)
    gm = np.load(calib_folder / "gelmap.npy").astype(F32)
    gel = gaussian_blur_host(gm, sim.deform_final_sigma(gm.shape)) * sensor.pixmm
    t.gel_map_shift = float(gel.max())
    gel_full = (gel - t.gel_map_shift).astype(F32)
    t.gel_map = np.ascontiguousarray(resize_bilinear_aa_host(gel_full, (H, W)).astype(F32))

    # background (taxim_torch.py:92-94,414-430,136-137)
    f0 = np.load(calib_folder / "dataPack.npz", allow_pickle=True)["f0"] / 255
    f0 = np.ascontiguousarray(np.moveaxis(f0.astype(F32), -1, 0)[::-1])  # HWC BGR -> CHW RGB
    f0b = gaussian_blur_host(f0, sim.initial_frame_sigma(f0.shape[1:]))
    d_i = (f0b - f0).mean(0)
    fmp = sim.frame_mixing_percentage
    bg_proc = np.where((d_i < sim.diff_threshold)[None], fmp * f0b + (1 - fmp) * f0, f0).astype(F32)
    t.background_full = np.ascontiguousarray(resize_bilinear_aa_host(bg_proc, (ch, cw)).astype(F32))
    t.background = np.ascontiguousarray(resize_bilinear_aa_host(bg_proc, (H, W)).astype(F32))

    # polynomial features live in calibration pixel units (taxim_torch.py:139-157)
    t.feat_x = np.ascontiguousarray(torch_linspace_f32(0, cw, W + 1)[:-1])
    t.feat_y = np.ascontiguousarray(torch_linspace_f32(0, ch, H + 1)[:-1])
    return t


def build_shadow_tables(calib_folder: Path, tables: TaximTables) -> dict:
    """Shadow calibration (taxim_torch.py:96-126) + the per-shape parameters of the shadow branch (taxim_torch.py:260-346)."""
    sim, H, W = tables.sim_params, tables.height, tables.width
    sd = np.load(Path(calib_folder) / "shadowTable.npz", allow_pickle=True)
    direction = sd["shadowDirections"].astype(F32)
    n_fan = int(sim.fan_angle * 2 / sim.fan_precision)
    fan = (direction[:, None] + np.linspace(-sim.fan_angle, sim.fan_angle, n_fan).astype(F32)[None, :]).astype(F32)
    tab = sd["shadowTable"][::-1]  # BGR -> RGB (taxim_torch.py:113); the "extra empty entry" concat keeps 24 heights
    nstep = max(len(e) for e in tab.reshape(-1))
    table = (np.array([list(e) + [np.inf] * (nstep - len(e)) for e in tab.reshape(-1)], dtype=F32)
             .reshape(tab.shape + (nstep,)) / F32(255))
    # two box-dilation rounds with conv2d(padding="same") (taxim_torch.py:261-272): composite window
    ks = np.array(sim.shadow_attachment_kernel_size((H, W)))  # (w, h)
    total = np.round(ks * 2).astype(np.int_)
    first = total // 2
    rounds = [np.maximum(1, first), np.maximum(1, total - first)]
    wl = sum(int((r[0] - 1) // 2) for r in rounds)
    wr = sum(int(r[0] - 1 - (r[0] - 1) // 2) for r in rounds)
    wt = sum(int((r[1] - 1) // 2) for r in rounds)
    wb = sum(int(r[1] - 1 - (r[1] - 1) // 2) for r in rounds)
    step_w, step_h = sim.shadow_step((H, W))
    sbw, sbh = sim.shadow_blur_sigma((H, W))
    kw, kh = gaussian_kernel_size(sbw), gaussian_kernel_size(sbh)
    # cos / sin of the fan angles, float32 like torch.cos / torch.sin of the float32 table (taxim_torch.py:299,303): computed
    # HERE so that device and CPU oracle multiply by the same bits - the ray sample coordinates are then integer-exact
    return dict(fan=np.ascontiguousarray(fan), fan_cos=np.ascontiguousarray(np.cos(fan), dtype=F32),
                fan_sin=np.ascontiguousarray(np.sin(fan), dtype=F32),
                table=np.ascontiguousarray(table), ndir=int(fan.shape[0]), nfan=int(n_fan),
                nheight=int(table.shape[2]), nstep=int(nstep), win=(wl, wr, wt, wb), depth0=0.4,
                height_precision=float(sim.height_precision), discretize_precision=float(sim.discretize_precision),
                step_x=float(step_h), step_y=float(step_w),  # (sic) x uses shadow_step[1] = the height-scaled value
                blur_kw=kw, blur_kh=kh, blur_taps_w=gaussian_taps(sbw, kw), blur_taps_h=gaussian_taps(sbh, kh))
