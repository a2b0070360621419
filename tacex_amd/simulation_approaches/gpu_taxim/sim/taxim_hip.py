"""`TaximHip` - the MI355X backend of the Taxim optical simulator.

Counterpart of the reference's `TaximTorch` (gpu_taxim/sim/taxim_torch.py:47-503) with the same public
surface (`render_direct`, `render`, `background_img`, `width`, `height`, `sim_params`, `sensor_params`,
`device`, `backend_name`, taxim_impl.py:74-246).  All per-frame arithmetic runs in hand-written HIP
kernels behind the C ABI of libtacex_hip.so; this class only prepares tables once per resolution,
owns the scratch buffers and passes raw device pointers + the current HIP stream.

There is no CPU fallback: constructing it without a visible AMD GPU raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Any

import numpy as np
import torch

from .... import _lib
from ....calibration import CALIB_GELSIGHT, TaximTables, build_shadow_tables, build_taxim_tables, load_params


class _ShapeCtx:
    """Device context + host tables for one tactile resolution."""

    def __init__(self, tables: TaximTables, device_index: int):
        lib = _lib.load_library()
        self.tables = tables
        p = _lib.TaximParams()
        p.height, p.width = tables.height, tables.width
        p.calib_height, p.calib_width = tables.sensor_params.height, tables.sensor_params.width
        p.pixmm = tables.sensor_params.pixmm
        p.num_bins = tables.sensor_params.num_bins
        p.contact_scale = tables.sim_params.contact_scale
        n = len(tables.ksize_w)
        if n > _lib.MAX_LEVELS:
            raise ValueError(f"at most {_lib.MAX_LEVELS} blur levels are supported, got {n}")
        p.n_levels = n
        self._keep = []  # keep the numpy buffers alive during the create call

        def fp(a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._keep.append(a)
            return a.ctypes.data_as(_lib.c_float_p)

        for i in range(n):
            p.ksize_w[i], p.ksize_h[i] = tables.ksize_w[i], tables.ksize_h[i]
            p.taps_w[i], p.taps_h[i] = fp(tables.taps_w[i]), fp(tables.taps_h[i])
        p.poly, p.gel_map, p.background = fp(tables.poly), fp(tables.gel_map), fp(tables.background)
        p.feat_x, p.feat_y = fp(tables.feat_x), fp(tables.feat_y)
        h = C.c_void_p()
        _lib.check(lib.tacex_taxim_create(device_index, C.byref(p), C.byref(h)), "tacex_taxim_create")
        self.handle = h
        self._lib = lib
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.tacex_taxim_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class TaximHip:
    def __init__(
        self,
        calib_folder: Path = CALIB_GELSIGHT,
        params: dict[str, dict[str, Any]] | None = None,
        device: torch.device | str = "cuda",
    ):
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise _lib.TacexHipError(
                f"TaximHip runs on an AMD GPU only (got device '{device}'); the reference's CPU path is not "
                "part of this package (the CPU oracle lives under oracle/ and is test infrastructure)."
            )
        self._device_index = self._device.index if self._device.index is not None else torch.cuda.current_device()
        self._device = torch.device("cuda", self._device_index)
        self._arch = _lib.require_gpu(self._device_index)
        self._calib_folder = Path(calib_folder)
        self._params_override = params
        self._sim_params, self._sensor_params = load_params(self._calib_folder, params)
        self._ctx: dict[tuple[int, int], _ShapeCtx] = {}
        self._ws: dict[tuple[int, int], torch.Tensor] = {}
        self._fmin: dict[tuple[int, int], torch.Tensor] = {}
        self._obs_scratch: dict[tuple[int, int], torch.Tensor] = {}
        self._frame_rows_key: dict[tuple[int, int], tuple[int, int]] = {}
        self._bg_full = None
        self._lib = _lib.load_library()

    # -- taxim_impl.py:204-246 ---------------------------------------------------------------------------
    @property
    def width(self) -> int:
        return self._sensor_params.width

    @property
    def height(self) -> int:
        return self._sensor_params.height

    @property
    def sim_params(self):
        return self._sim_params

    @property
    def sensor_params(self):
        return self._sensor_params

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def backend_name(self) -> str:
        return "hip"

    # -- tables ------------------------------------------------------------------------------------------
    def context(self, shape_hw: tuple[int, int]) -> _ShapeCtx:
        shape_hw = (int(shape_hw[0]), int(shape_hw[1]))
        ctx = self._ctx.get(shape_hw)
        if ctx is None:
            tables = build_taxim_tables(self._calib_folder, shape_hw, self._params_override)
            ctx = _ShapeCtx(tables, self._device_index)
            self._ctx[shape_hw] = ctx
        return ctx

    @property
    def background_img(self) -> torch.Tensor:
        """(3, calib_h, calib_w) processed background, like taxim_torch.py:132-134."""
        if self._bg_full is None:
            t = self.context((self.height, self.width)).tables
            self._bg_full = torch.from_numpy(t.background_full).to(self._device)
        return self._bg_full

    def background_for(self, shape_hw: tuple[int, int]) -> torch.Tensor:
        return torch.from_numpy(self.context(shape_hw).tables.background).to(self._device)

    def gel_map_for(self, shape_hw: tuple[int, int]) -> torch.Tensor:
        return torch.from_numpy(self.context(shape_hw).tables.gel_map).to(self._device)

    # -- scratch -----------------------------------------------------------------------------------------
    def _ensure_shadow(self, ctx: _ShapeCtx):
        """Upload the shadow-branch tables of this resolution once (taxim_torch.py:96-126, 260-346)."""
        if getattr(ctx, "shadow_ready", False):
            return
        sh = build_shadow_tables(self._calib_folder, ctx.tables)
        p = _lib.ShadowParams()
        p.num_directions, p.num_fan_rays, p.num_heights, p.num_steps = sh["ndir"], sh["nfan"], sh["nheight"], sh["nstep"]
        keep = [np.ascontiguousarray(sh[k], dtype=np.float32) for k in ("fan", "table", "blur_taps_w", "blur_taps_h", "fan_cos", "fan_sin")]
        p.fan_angles, p.table = keep[0].ctypes.data_as(_lib.c_float_p), keep[1].ctypes.data_as(_lib.c_float_p)
        p.fan_cos, p.fan_sin = keep[4].ctypes.data_as(_lib.c_float_p), keep[5].ctypes.data_as(_lib.c_float_p)
        p.win_left, p.win_right, p.win_top, p.win_bottom = sh["win"]
        p.shadow_depth_0, p.height_precision, p.discretize_precision = sh["depth0"], sh["height_precision"], sh["discretize_precision"]
        p.step_x, p.step_y = sh["step_x"], sh["step_y"]
        p.blur_kw, p.blur_kh = sh["blur_kw"], sh["blur_kh"]
        p.blur_taps_w, p.blur_taps_h = keep[2].ctypes.data_as(_lib.c_float_p), keep[3].ctypes.data_as(_lib.c_float_p)
        _lib.check(self._lib.tacex_taxim_set_shadow(ctx.handle, C.byref(p)), "tacex_taxim_set_shadow")
        ctx.shadow_ready = True

    def _workspace(self, ctx: _ShapeCtx, shape_hw, B: int, with_shadow: bool = False) -> torch.Tensor:
        need = self._lib.tacex_taxim_workspace_bytes(ctx.handle, B)
        if with_shadow:
            need += self._lib.tacex_taxim_shadow_workspace_bytes(ctx.handle, B)
        ws = self._ws.get(shape_hw)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self._device)
            self._ws[shape_hw] = ws
        return ws

    def _frame_min_buf(self, shape_hw, B: int) -> torch.Tensor:
        fm = self._fmin.get(shape_hw)
        if fm is None or fm.numel() < B:
            fm = torch.empty(B, dtype=torch.float32, device=self._device)
            self._fmin[shape_hw] = fm
        return fm[:B]

    def _check_hm(self, height_map: torch.Tensor) -> torch.Tensor:
        if not isinstance(height_map, torch.Tensor):
            raise TypeError("height_map must be a torch.Tensor on the GPU (use render() for NumPy input)")
        if height_map.device != self._device:
            raise ValueError(f"height_map is on {height_map.device}, this simulator runs on {self._device}")
        if height_map.dim() < 2:
            raise ValueError("height_map needs at least 2 dimensions (..., H, W)")
        hm = height_map.reshape((-1,) + tuple(height_map.shape[-2:]))
        if hm.dtype != torch.float32:
            hm = hm.float()
        return hm.contiguous()

    def _press(self, press_depth, B: int):
        if press_depth is None:
            return None
        if not isinstance(press_depth, torch.Tensor):
            press_depth = torch.full((B,), float(press_depth), dtype=torch.float32, device=self._device)
        p = press_depth.to(device=self._device, dtype=torch.float32).reshape(-1)
        if p.numel() == 1 and B != 1:
            p = p.expand(B)
        if p.numel() != B:
            raise ValueError(f"press_depth has {p.numel()} entries for {B} height maps")
        return p.contiguous()

    # -- taxim_impl.py:153-163 / taxim_torch.py:174-195 ----------------------------------------------------
    def render_direct(
        self,
        height_map: torch.Tensor,
        with_shadow: bool = True,
        press_depth: torch.Tensor | float | None = None,
        orig_hm_fmt: bool = False,
        out: torch.Tensor | None = None,
        frame_min: torch.Tensor | None = None,
        z_out: torch.Tensor | None = None,
        mask_out: torch.Tensor | None = None,
        obs_out: torch.Tensor | None = None,
        frame_rows: torch.Tensor | None = None,
    ) -> torch.Tensor:
        """(..., H, W) mm height map -> (..., 3, H, W) RGB in [0,1] (a channel-first VIEW of an NHWC buffer).

        `obs_out` (B, oh, ow, 3) float32 or uint8 (= floor(255 x + 0.5)): additionally produce the antialiased low-resolution policy observation in the same
        pass (fused into the tail kernel where one exists).

        Extra keyword arguments (not in the reference): `out` (B,H,W,3) buffer to render into,
        `frame_min` (B,) precomputed per-frame minimum (+ `frame_rows` (B,4) int32, the contact row and column ranges produced with it),
        `z_out` / `mask_out` to also return the deformed gel
        and the shrunken contact mask of taxim_torch.py:443-473 (the FOTS wrapper needs both).
        """
        batch_shape = tuple(height_map.shape[:-2])
        hm = self._check_hm(height_map)
        B, H, W = hm.shape
        if B == 0:  # empty batch: nothing to launch
            empty = torch.empty((0, H, W, 3), dtype=torch.float32, device=self._device) if out is None else out
            return empty.movedim(3, 1).reshape(batch_shape + (3, H, W))
        ctx = self.context((H, W))
        if orig_hm_fmt:  # taxim_torch.py:185-186
            hm = ctx.tables.gel_map_shift - hm
        press = self._press(press_depth, B)
        flags = 0
        if press is None:
            flags |= _lib.FLAG_NO_SHIFT
        if frame_min is not None:
            flags |= _lib.FLAG_HAVE_FRAME_MIN
            fmin = frame_min
            if frame_rows is not None and frame_rows.dtype == torch.int32 and frame_rows.is_contiguous() and frame_rows.shape[0] >= B \
                    and frame_rows.dim() == 2 and frame_rows.shape[1] == 4:
                key = (frame_rows.data_ptr(), int(frame_rows.shape[0]))
                if self._frame_rows_key.get((H, W)) != key:  # registered once per buffer
                    _lib.check(self._lib.tacex_taxim_set_frame_rows(ctx.handle, _lib.ptr(frame_rows), int(frame_rows.shape[0])),
                               "tacex_taxim_set_frame_rows")
                    self._frame_rows_key[(H, W)] = key
                flags |= _lib.FLAG_HAVE_FRAME_ROWS
        else:
            fmin = self._frame_min_buf((H, W), B)
        if out is None:
            out = torch.empty((B, H, W, 3), dtype=torch.float32, device=self._device)
        elif tuple(out.shape) != (B, H, W, 3) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("out must be a contiguous float32 (B,H,W,3) tensor")
        if with_shadow:
            self._ensure_shadow(ctx)
            flags |= _lib.FLAG_WITH_SHADOW
        ws = self._workspace(ctx, (H, W), B, with_shadow)
        with torch.cuda.device(self._device):
            stream = _lib.current_stream_handle(self._device)
            if obs_out is not None:
                if obs_out.dim() != 4 or obs_out.shape[0] != B or obs_out.shape[3] != 3 or not obs_out.is_contiguous():
                    raise ValueError("obs_out must be a contiguous (B, oh, ow, 3) tensor")
                if obs_out.dtype not in (torch.float32, torch.uint8):
                    raise ValueError("obs_out must be float32 or uint8")
                oh, ow = int(obs_out.shape[1]), int(obs_out.shape[2])
                need = B * (max(H * ow, oh * W) + oh * ow) * 3
                if obs_out.dtype == torch.uint8:
                    flags |= _lib.FLAG_OBS_U8
                sc = self._obs_scratch.get((H, W))
                if sc is None or sc.numel() < need:
                    sc = torch.empty(need, dtype=torch.float32, device=self._device)
                    self._obs_scratch[(H, W)] = sc
                rc = self._lib.tacex_taxim_render_obs(
                    ctx.handle, _lib.ptr(hm), _lib.ptr(press), _lib.ptr(fmin), _lib.ptr(out), _lib.ptr(z_out),
                    _lib.ptr(mask_out), _lib.ptr(ws), _lib.ptr(sc), _lib.ptr(obs_out), oh, ow, B, flags, stream)
                _lib.check(rc, "tacex_taxim_render_obs")
            else:
                rc = self._lib.tacex_taxim_render(
                    ctx.handle, _lib.ptr(hm), _lib.ptr(press), _lib.ptr(fmin), _lib.ptr(out), _lib.ptr(z_out),
                    _lib.ptr(mask_out), _lib.ptr(ws), B, flags, stream)
                _lib.check(rc, "tacex_taxim_render")
        return out.movedim(3, 1).reshape(batch_shape + (3, H, W))

    def deform(self, height_map: torch.Tensor, press_depth, frame_min: torch.Tensor | None = None,
               z_out: torch.Tensor | None = None, mask_out: torch.Tensor | None = None):
        """__get_shifted_height_map + __compute_gel_pad_deformation (taxim_torch.py:432-473):
        returns (deformed gel (B,H,W) f32 mm, shrunken contact mask (B,H,W) uint8)."""
        hm = self._check_hm(height_map)
        B, H, W = hm.shape
        ctx = self.context((H, W))
        press = self._press(press_depth, B)
        flags = 0 if press is not None else _lib.FLAG_NO_SHIFT
        if frame_min is not None:
            flags |= _lib.FLAG_HAVE_FRAME_MIN
            fmin = frame_min
        else:
            fmin = self._frame_min_buf((H, W), B)
        if z_out is None:
            z_out = torch.empty((B, H, W), dtype=torch.float32, device=self._device)
        if mask_out is None:
            mask_out = torch.empty((B, H, W), dtype=torch.uint8, device=self._device)
        ws = self._workspace(ctx, (H, W), B)
        with torch.cuda.device(self._device):
            rc = self._lib.tacex_taxim_deform(
                ctx.handle, _lib.ptr(hm), _lib.ptr(press), _lib.ptr(fmin), _lib.ptr(z_out), _lib.ptr(mask_out),
                _lib.ptr(ws), B, flags, _lib.current_stream_handle(self._device))
        _lib.check(rc, "tacex_taxim_deform")
        return z_out, mask_out

    def shadow_rays(self, deformed_gel: torch.Tensor, contact_mask: torch.Tensor, grad_dir: torch.Tensor) -> torch.Tensor:
        """Ray march of the shadow branch alone (taxim_torch.py:261-337): (B,H,W) deformed gel [mm], uint8 shrunken contact
        mask and gradient direction -> (B,H,W,3) per-pixel / channel minimum of the shadow-table samples (+inf: none)."""
        z = self._check_hm(deformed_gel)
        B, H, W = z.shape
        ctx = self.context((H, W))
        self._ensure_shadow(ctx)
        m = contact_mask.to(self._device, torch.uint8).reshape(B, H, W).contiguous()
        g = grad_dir.to(self._device, torch.float32).reshape(B, H, W).contiguous()
        out = torch.empty((B, H, W, 3), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            rc = self._lib.tacex_taxim_shadow_rays(ctx.handle, _lib.ptr(z), _lib.ptr(m), _lib.ptr(g), _lib.ptr(out), B,
                                                   _lib.current_stream_handle(self._device))
        _lib.check(rc, "tacex_taxim_shadow_rays")
        return out

    def shade(self, deformed_gel: torch.Tensor, return_bins: bool = False):
        """taxim_torch.py:237-258 on an existing deformed gel: (B,H,W) -> (B,H,W,3) [+ (B,H,W,2) uint8 bins]."""
        z = self._check_hm(deformed_gel)
        B, H, W = z.shape
        ctx = self.context((H, W))
        rgb = torch.empty((B, H, W, 3), dtype=torch.float32, device=self._device)
        idx = torch.empty((B, H, W, 2), dtype=torch.uint8, device=self._device) if return_bins else None
        with torch.cuda.device(self._device):
            rc = self._lib.tacex_taxim_shade(ctx.handle, _lib.ptr(z), _lib.ptr(rgb), _lib.ptr(idx), B,
                                             _lib.current_stream_handle(self._device))
        _lib.check(rc, "tacex_taxim_shade")
        return (rgb, idx) if return_bins else rgb

    # -- taxim_impl.py:117-151, taxim_torch.py:166-171 ------------------------------------------------------
    def convert_height_map(self, height_map: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.asarray(height_map)).to(self._device).float()

    def img_to_numpy(self, img: torch.Tensor) -> np.ndarray:
        b_dims = len(img.shape[:-3])
        return img.permute(*range(b_dims), -2, -1, -3).cpu().numpy()

    def render(self, height_map, with_shadow: bool = True, press_depth=None, orig_hm_fmt: bool = False):
        """NumPy in -> NumPy (…,H,W,3) out; tensors are passed straight to render_direct."""
        if isinstance(height_map, np.ndarray):
            pd = press_depth
            if isinstance(pd, np.ndarray):
                pd = torch.from_numpy(pd).to(self._device).float()
            res = self.render_direct(self.convert_height_map(height_map), with_shadow, pd, orig_hm_fmt)
            return self.img_to_numpy(res)
        return self.render_direct(height_map, with_shadow, press_depth, orig_hm_fmt)

    __call__ = render

    def set_fused_tail(self, shape_hw, enabled):
        """Ablation hook: 0 / False = every pyramid level as its own kernel + separate shade; 1 / True = fused tail (default:
        streaming kernel for plain renders, LDS-tiled kernel when full deformed-gel / mask frames are requested); 2 = always the
        LDS-tiled kernel."""
        ctx = self.context(shape_hw)
        _lib.check(self._lib.tacex_taxim_set_fused_tail(ctx.handle, int(enabled)), "set_fused_tail")

    # -- FOTS contact statistics as a by-product of the render (fused tail only) -------------------------------------
    def fots_partials_per_env(self, shape_hw) -> int:
        """Records per env the fused tail writes (0: no fused tail for this shape / tail disabled)."""
        return int(self._lib.tacex_taxim_fots_partials_per_env(self.context(shape_hw).handle))

    def set_fots_partials(self, shape_hw, buf: torch.Tensor | None, capacity_frames: int = 0):
        """buf: uint8 tensor of capacity_frames * fots_partials_per_env * 16 bytes (or None to disable)."""
        ctx = self.context(shape_hw)
        _lib.check(self._lib.tacex_taxim_set_fots_partials(ctx.handle, _lib.ptr(buf) if buf is not None else 0,
                                                           int(capacity_frames)), "set_fots_partials")

    def set_fots_taps(self, shape_hw, marker_x, marker_y, z_pix: torch.Tensor | None, mask_pix: torch.Tensor | None,
                      capacity_frames: int = 0):
        """marker_x / marker_y: host int32 arrays (M,); z_pix (cap, M) f32 / mask_pix (cap, M) u8 device tensors the fused
        tail fills with the deformed gel / contact mask at the marker pixels (None disables)."""
        import numpy as np

        ctx = self.context(shape_hw)
        if z_pix is None:
            _lib.check(self._lib.tacex_taxim_set_fots_taps(ctx.handle, 0, 0, 0, 0, 0, 0), "set_fots_taps")
            return
        mx = np.ascontiguousarray(marker_x, dtype=np.int32)
        my = np.ascontiguousarray(marker_y, dtype=np.int32)
        _lib.check(self._lib.tacex_taxim_set_fots_taps(ctx.handle, mx.ctypes.data, my.ctypes.data, int(mx.size), _lib.ptr(z_pix),
                                                       _lib.ptr(mask_pix), int(capacity_frames)), "set_fots_taps")

    def chunk_frames(self, shape_hw, num_frames: int) -> int:
        """Frames per pass of the kernel sequence for a `num_frames` call (Infinity-Cache-sized chunks)."""
        return int(self._lib.tacex_taxim_chunk_frames(self.context(shape_hw).handle, int(num_frames)))

    # -- profiling (bench.py roofline leg) --------------------------------------------------------------------
    def set_profiling(self, shape_hw, enabled: bool):
        ctx = self.context(shape_hw)
        _lib.check(self._lib.tacex_taxim_set_profiling(ctx.handle, 1 if enabled else 0), "set_profiling")

    def read_profile(self, shape_hw) -> dict[str, tuple[float, int]]:
        ctx = self.context(shape_hw)
        out = {}
        for s in range(self._lib.tacex_taxim_num_stages(ctx.handle)):
            ms, n = C.c_double(0), C.c_int(0)
            _lib.check(self._lib.tacex_taxim_read_profile(ctx.handle, s, C.byref(ms), C.byref(n)), "read_profile")
            out[self._lib.tacex_taxim_stage_name(ctx.handle, s).decode()] = (ms.value, n.value)
        return out
