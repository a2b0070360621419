"""`FOTSMarkerSimulator` - marker-motion plugin, drop-in for the reference's
source/tacex/tacex/simulation_approaches/fots/fots_marker_sim.py:26-184.

The reference re-runs the Taxim deformation, then loops over envs in Python with `.cpu()` syncs and a
NumPy `MarkerMotion` per env (fots_marker_sim.py:132-182).  Here the deformed gel / contact mask are
shared with the optical simulator when it rendered the same height map, and all envs are handled by two
HIP launches (csrc/fots_kernels.hip).  The IsaacLab `FrameTransformer` that supplies the indenter's yaw
(fots_marker_sim.py:147-159) is replaced by `cfg.yaw_source` / `set_indenter_yaw`.
"""
from __future__ import annotations

import ctypes as C
from typing import TYPE_CHECKING

import numpy as np
import torch

from ... import _lib
from ..gelsight_simulator import GelSightSimulator
from ..gpu_taxim.taxim_sim import TaximSimulator

if TYPE_CHECKING:
    from ...gelsight_sensor import GelSightSensor
    from .fots_marker_sim_cfg import FOTSMarkerSimulatorCfg

FOTS_LAMB = (0.00125, 0.00021, 0.00038)  # hard-coded in the reference (fots_marker_sim.py:77); cfg.lamb is ignored


def marker_grid(width: int, height: int, num_markers_col: int, num_markers_row: int, x0, y0):
    """Initial marker pixels (marker_motion.py:59-76): integer linspace (truncating), row-major (row, col)."""
    xi = np.linspace(x0, width - x0, num_markers_col, dtype=int)
    yi = np.linspace(y0, height - y0, num_markers_row, dtype=int)
    gx, gy = np.meshgrid(xi, yi)
    return gx.reshape(-1).astype(np.int32), gy.reshape(-1).astype(np.int32)


class FOTSMarkerSimulator(GelSightSimulator):
    cfg: "FOTSMarkerSimulatorCfg"

    def __init__(self, sensor: "GelSightSensor", cfg: "FOTSMarkerSimulatorCfg"):
        self.sensor = sensor
        super().__init__(sensor=sensor, cfg=cfg)
        self._handle = None
        self._theta = None

    def _initialize_impl(self):
        self._device = self.sensor.device if self.cfg.device is None else self.cfg.device
        self._num_envs = self.sensor._num_envs
        self._indentation_depth = torch.zeros((self._num_envs,), device=self.sensor._device)
        # FOTS needs the Taxim deformation (fots_marker_sim.py:58-66)
        if (self.sensor.optical_simulator is not None) and (type(self.sensor.optical_simulator) is TaximSimulator):
            self._optical: TaximSimulator = self.sensor.optical_simulator
            if not hasattr(self._optical, "_taxim"):
                self._optical._initialize_impl()
            self._taxim = self._optical._taxim
        else:
            raise RuntimeError(
                "Currently FOTS simulation approach has to be used in combination with GPU-Taxim as optical-simulator."
            )
        W, H = self.cfg.tactile_img_res
        mp = self.cfg.marker_params
        mx, my = marker_grid(W, H, mp.num_markers_col, mp.num_markers_row, mp.x0, mp.y0)
        if mx.size != mp.num_markers:
            raise RuntimeError(f"marker_params.num_markers={mp.num_markers} != rows*cols={mx.size}")
        self.init_marker_pos = np.stack((mx, my), axis=-1).astype(np.int64)  # (M, 2) [x, y]
        self.img_res = self.cfg.tactile_img_res

        lib = _lib.load_library()
        p = _lib.FotsParams()
        p.height, p.width = H, W
        p.num_markers_row, p.num_markers_col = mp.num_markers_row, mp.num_markers_col
        p.marker_x = mx.ctypes.data_as(_lib.c_int32_p)
        p.marker_y = my.ctypes.data_as(_lib.c_int32_p)
        for k in range(3):
            p.lamb[k] = FOTS_LAMB[k]
        p.mm2pix = self.cfg.mm_to_pixel
        p.shear_max = 10.0
        p.theta_max_deg = 60.0
        dev = torch.device(self._device)
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        _lib.check(lib.tacex_fots_create(self._dev_index, C.byref(p), C.byref(h)), "tacex_fots_create")
        self._handle = h
        self._lib = lib

        B, M = self._num_envs, mp.num_markers
        # marker flow, (num_envs, 2, num_markers, 2): dim1 = [initial, current], dim3 = [x, y] (fots_marker_sim.py:90-99)
        self.marker_data = torch.zeros((B, 2, M, 2), device=self._device)
        self.marker_data[:, 0] = torch.tensor(self.init_marker_pos, device=self._device, dtype=torch.float32)
        self.marker_data[:, 1] = self.marker_data[:, 0]
        # trajectory state replaces output["traj"] lists (only traj[0], traj[-1], len are ever read, marker_motion.py:177-205)
        self._traj_state = torch.zeros((B, 8), device=self._device)
        self.sensor._data.output["traj"] = self._traj_state
        self.theta = torch.zeros((B,), device=self._device)
        self._ws = torch.empty(max(1, lib.tacex_fots_workspace_bytes(B)), dtype=torch.uint8, device=self._device)
        self._optical.request_deformation_outputs(mx, my) if tuple(self._optical.cfg.tactile_img_res) == tuple(self.cfg.tactile_img_res) \
            else self._optical.request_deformation_outputs()
        self._z = None
        self._mask = None

    def __del__(self):
        try:
            if self._handle:
                self._lib.tacex_fots_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def set_indenter_yaw(self, theta: torch.Tensor):
        """Yaw [rad] of the indenter in the sensor frame per env (what the FrameTransformer yields, fots_marker_sim.py:155-158)."""
        self.theta = theta.to(device=self._device, dtype=torch.float32).reshape(-1).contiguous()

    def marker_motion_simulation(self):
        self._indentation_depth = self.sensor._indentation_depth
        if self.cfg.yaw_source is not None:
            self.set_indenter_yaw(self.cfg.yaw_source())
        opt = self._optical
        partials = None
        if getattr(opt, "_fots_compact_version", -1) == self.sensor._height_map_version \
                and getattr(opt, "_fots_partials_version", -1) == self.sensor._height_map_version:
            # the render left the marker-pixel values and the contact statistics behind: no frame is read at all
            indent = self._indentation_depth.to(self._device).contiguous()
            with torch.cuda.device(self.marker_data.device):
                rc = self._lib.tacex_fots_markers_compact(
                    self._handle, _lib.ptr(opt._pix_z), _lib.ptr(opt._pix_m), _lib.ptr(indent), _lib.ptr(self.theta),
                    _lib.ptr(self._traj_state), _lib.ptr(self.marker_data), _lib.ptr(self._ws), _lib.ptr(opt._fots_partials),
                    int(opt._fots_partials.shape[1]), self._num_envs, _lib.current_stream_handle(self.marker_data.device))
            _lib.check(rc, "tacex_fots_markers_compact")
            return self.marker_data
        if opt._keep_deformation and opt._deformation_version == self.sensor._height_map_version \
                and tuple(opt.cfg.tactile_img_res) == tuple(self.cfg.tactile_img_res):
            z, mask = opt._deformed_gel, opt._contact_mask  # same height map already deformed by the render
            if getattr(opt, "_fots_partials_version", -1) == self.sensor._height_map_version:
                partials = opt._fots_partials  # ... and its contact statistics came out of the same kernel
        else:
            height_map = self.sensor._data.output["height_map"]
            W, H = self.cfg.tactile_img_res
            if (height_map.shape[1], height_map.shape[2]) != (H, W):  # fots_marker_sim.py:121-122
                resized = torch.empty((height_map.shape[0], H, W), device=self._device)
                with torch.cuda.device(resized.device):
                    rc = self._lib.tacex_resize_bilinear_aa(
                        _lib.ptr(height_map.contiguous()), height_map.shape[1], height_map.shape[2], _lib.ptr(resized),
                        H, W, height_map.shape[0], _lib.current_stream_handle(resized.device))
                _lib.check(rc, "tacex_resize_bilinear_aa")
                height_map = resized
            if self._z is None:
                self._z = torch.empty((self._num_envs, H, W), device=self._device)
                self._mask = torch.empty((self._num_envs, H, W), dtype=torch.uint8, device=self._device)
            z, mask = self._taxim.deform(height_map, self._indentation_depth, z_out=self._z, mask_out=self._mask)
        indent = self._indentation_depth.to(self._device).contiguous()
        with torch.cuda.device(self.marker_data.device):
            if partials is not None:
                rc = self._lib.tacex_fots_markers_partials(
                    self._handle, _lib.ptr(z), _lib.ptr(mask), _lib.ptr(indent), _lib.ptr(self.theta),
                    _lib.ptr(self._traj_state), _lib.ptr(self.marker_data), _lib.ptr(self._ws), _lib.ptr(partials),
                    int(partials.shape[1]), self._num_envs, _lib.current_stream_handle(self.marker_data.device))
            else:
                rc = self._lib.tacex_fots_markers(
                    self._handle, _lib.ptr(z), _lib.ptr(mask), _lib.ptr(indent), _lib.ptr(self.theta),
                    _lib.ptr(self._traj_state), _lib.ptr(self.marker_data), _lib.ptr(self._ws), self._num_envs,
                    _lib.current_stream_handle(self.marker_data.device))
        _lib.check(rc, "tacex_fots_markers")
        return self.marker_data

    # -- marker image (fots_marker_sim.py:346-384, 265-272) --------------------------------------------------
    def set_patch_array(self, patch_array_dict: dict):
        """The pre-drawn marker patches (`generate_patch_array()` of the reference, FS:387-446, or `marker_patches.load_patch_array`)."""
        from .marker_patches import check_patch_array

        check_patch_array(patch_array_dict)
        self.patch_array_dict = patch_array_dict
        self._patch_dev = torch.from_numpy(np.ascontiguousarray(patch_array_dict["patch_array"])).to(self._device)

    def _patches(self):
        if getattr(self, "_patch_dev", None) is None:
            from .marker_patches import generate_patch_array

            self.set_patch_array(generate_patch_array())  # NumPy stand-in for the OpenCV-drawn table (see marker_patches.py)
        return self._patch_dev

    def marker_images(self, marker_data: torch.Tensor | None = None, marker_size: float = 3, overlay_rgb: torch.Tensor | None = None,
                      img_res: tuple | None = None):
        """Marker image of EVERY env in one launch: (B, H, W) uint8, white canvas with the dot patch of each marker stamped at
        its current position (`draw_markers`, FS:346-384).  With `overlay_rgb` (B, H, W, 3) float32 in [0,1] also the RGB x marker
        overlay the reference shows (FS:265-272): uint8(rgb * 255 * marker / 255).  Returns (images, overlay | None)."""
        md = self.marker_data if marker_data is None else marker_data
        md = md.to(self._device, torch.float32).contiguous()
        B, _, M, _ = md.shape
        W, H = self.cfg.tactile_img_res if img_res is None else img_res
        pt = self._patches()
        d = self.patch_array_dict
        sr, S = int(d["super_resolution_ratio"]), int(d["size_slot_num"])
        import math

        pw = math.floor((marker_size - d["base_circle_radius"]) * sr)  # FS:370-373
        img = torch.empty((B, H, W), dtype=torch.uint8, device=self._device)
        ov = None
        if overlay_rgb is not None:
            if tuple(overlay_rgb.shape) != (B, H, W, 3) or overlay_rgb.dtype != torch.float32:
                raise ValueError(f"overlay_rgb must be float32 ({B}, {H}, {W}, 3)")
            overlay_rgb = overlay_rgb.contiguous()
            ov = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self._device)
        with torch.cuda.device(img.device):
            rc = self._lib.tacex_fots_marker_image(_lib.ptr(md), _lib.ptr(pt), sr, S, int(pw), _lib.ptr(overlay_rgb), _lib.ptr(img),
                                                   _lib.ptr(ov), B, M, H, W, _lib.current_stream_handle(img.device))
        _lib.check(rc, "tacex_fots_marker_image")
        return img, ov

    def draw_markers(self, marker_uv: np.ndarray, marker_size=3, img_w=320, img_h=240) -> np.ndarray:
        """Reference signature (FS:346): marker positions (num_markers, 2) of ONE sensor -> (img_h, img_w) uint8."""
        uv = torch.from_numpy(np.asarray(marker_uv, dtype=np.float32)).to(self._device)
        md = torch.stack((uv, uv), 0)[None]
        img, _ = self.marker_images(md, marker_size, img_res=(img_w, img_h))
        return img[0].cpu().numpy()

    def reset(self):
        """fots_marker_sim.py:206-208.  The trajectories are NOT cleared here (the reference does not either): an env's
        trajectory restarts when the marker simulation sees its indentation depth at 0 (FS:176-177), which the sensor's
        reset guarantees for the envs being reset; envs still in contact keep their shear / twist origin."""
        self._indentation_depth = torch.zeros((self._num_envs,), device=self._device)

    def _set_debug_vis_impl(self, debug_vis: bool):
        pass

    def _debug_vis_callback(self, event):
        pass
