"""`TaximSimulator` - optical simulation plugin, drop-in for the reference's
source/tacex/tacex/simulation_approaches/gpu_taxim/taxim_sim.py:20-135 (same attributes and methods),
with the arithmetic done by libtacex_hip.so instead of PyTorch ops."""
from __future__ import annotations

from pathlib import Path
from typing import TYPE_CHECKING

import torch

from ... import _lib
from ..gelsight_simulator import GelSightSimulator
from .sim import Taxim

if TYPE_CHECKING:
    from ...gelsight_sensor import GelSightSensor
    from .taxim_sim_cfg import TaximSimulatorCfg


class TaximSimulator(GelSightSimulator):
    cfg: "TaximSimulatorCfg"

    def __init__(self, sensor: "GelSightSensor", cfg: "TaximSimulatorCfg"):
        self.sensor = sensor
        super().__init__(sensor=sensor, cfg=cfg)

    def _initialize_impl(self):
        calib_folder = Path(self.cfg.calib_folder_path)
        self._device = self.sensor.device if self.cfg.device is None else self.cfg.device
        self._num_envs = self.sensor._num_envs
        W, H = self.cfg.tactile_img_res
        # indentation depth in mm (taxim_sim.py:43-50), on the sensor's device like the reference
        self._indentation_depth = torch.zeros((self._num_envs,), device=self.sensor._device)
        self.tactile_rgb_img = torch.zeros((self._num_envs, H, W, 3), device=self._device)
        self._taxim = Taxim(calib_folder=calib_folder, device=self._device)
        # (3,H,W) -> (H,W,3); the reference resizes the calibration-size background (taxim_sim.py:64-72),
        # the tables already hold the same resize for the tactile resolution
        self.background_img = self._taxim.background_for((H, W)).movedim(0, 2).contiguous()
        self.tactile_rgb_img[:] = self.background_img
        self.img_res = self.cfg.tactile_img_res
        # per-frame minimum of the current height map, shared between compute_indentation_depth and the render
        self._frame_min = torch.zeros((self._num_envs,), device=self._device)
        self._frame_min_version = -1
        # first / last frame row and first / last frame column with a pixel below the press plane, produced with the minimum (tacex_height_map_from_depth /
        # tacex_indentation_depth); the render skips the pyramid bands that cannot be non-zero
        self._frame_rows = torch.zeros((self._num_envs, 4), dtype=torch.int32, device=self._device)  # [row lo, row hi, column lo, column hi]
        self._frame_rows_version = -1
        self._indent_version = -1
        # deformed gel + contact mask of the latest render, kept for a marker simulator (FOTS re-uses them)
        self._keep_deformation = False
        self._deformed_gel = None
        self._contact_mask = None
        self._deformation_version = -1
        self._fots_partials_version = -1
        self._fots_compact_version = -1
        self._resized_hm = None
        self.policy_obs = None
        if getattr(self.cfg, "policy_obs_res", None) is not None:
            ow, oh = self.cfg.policy_obs_res
            dt = {"float32": torch.float32, "uint8": torch.uint8}[getattr(self.cfg, "policy_obs_dtype", "float32")]
            self.policy_obs = torch.zeros((self._num_envs, oh, ow, 3), device=self._device, dtype=dt)

    # -- helpers --------------------------------------------------------------------------------------
    def request_deformation_outputs(self, marker_x=None, marker_y=None):
        """A marker simulator asks the optical simulator to also keep (deformed gel, contact mask).  With the marker
        pixel grid (host int arrays) and a fused tail, only the values AT the markers are kept (`_pix_z`, `_pix_m`,
        (num_envs, M)) and the 5 B/px full-frame stores are dropped; the full frames are the fallback."""
        W, H = self.cfg.tactile_img_res
        self._keep_deformation = True
        self._deformed_gel = torch.zeros((self._num_envs, H, W), device=self._device)
        self._contact_mask = torch.zeros((self._num_envs, H, W), dtype=torch.uint8, device=self._device)
        self._pix_z = self._pix_m = None
        self._fots_compact_version = -1
        if marker_x is not None and self._taxim.fots_partials_per_env((H, W)) > 0:
            M = int(len(marker_x))
            self._pix_z = torch.zeros((self._num_envs, M), device=self._device)
            self._pix_m = torch.zeros((self._num_envs, M), dtype=torch.uint8, device=self._device)
            self._taxim.set_fots_taps((H, W), marker_x, marker_y, self._pix_z, self._pix_m, self._num_envs)
        # the fused tail also leaves FOTS's per-env contact statistics behind (one 16-byte record per wave and tile)
        n = self._taxim.fots_partials_per_env((H, W))
        self._fots_partials = torch.zeros((self._num_envs, max(n, 1), 16), dtype=torch.uint8, device=self._device)
        self._fots_partials_version = -1
        if n > 0:
            self._taxim.set_fots_partials((H, W), self._fots_partials, self._num_envs)

    def _tactile_height_map(self) -> tuple[torch.Tensor, bool]:
        """Height map at the tactile resolution; resized with the HIP kernel if the camera differs (taxim_sim.py:88-89)."""
        height_map = self.sensor._data.output["height_map"]
        W, H = self.cfg.tactile_img_res
        if (height_map.shape[1], height_map.shape[2]) == (H, W):
            return height_map, False
        B = height_map.shape[0]
        if self._resized_hm is None or self._resized_hm.shape != (B, H, W):
            self._resized_hm = torch.empty((B, H, W), dtype=torch.float32, device=self._device)
        lib = _lib.load_library()
        hm = height_map.contiguous()
        with torch.cuda.device(self._resized_hm.device):
            rc = lib.tacex_resize_bilinear_aa(_lib.ptr(hm), hm.shape[1], hm.shape[2], _lib.ptr(self._resized_hm), H, W, B,
                                              _lib.current_stream_handle(self._resized_hm.device))
        _lib.check(rc, "tacex_resize_bilinear_aa")
        return self._resized_hm, True

    def defer_height_map_from_depth(self, depth, near, far, hm, fmin, indent, cam_u8, rows):
        """The sensor's depth -> height map pass (GS:581-593 + TS:115-131), handed to the NEXT `optical_simulation()` instead of being
        launched now: the render runs it chunk by chunk beside its band levels (`tacex_taxim_defer_height_map_from_depth`).  The caller
        (GelSightSensor._get_height_map) guarantees that the render follows in the same update and reads exactly these buffers."""
        B, H, W = hm.shape
        lib = _lib.load_library()
        handle = self._taxim.context((H, W)).handle
        with torch.cuda.device(hm.device):
            # (a pass left pending by an update that failed half way runs now, so that this one can be queued)
            _lib.check(lib.tacex_taxim_flush_deferred(handle, _lib.current_stream_handle(hm.device)), "tacex_taxim_flush_deferred")
            rc = lib.tacex_taxim_defer_height_map_from_depth(
                handle, _lib.ptr(depth), float(near), float(far), float(self.cfg.gelpad_height),
                float(self.cfg.gelpad_to_camera_min_distance), _lib.ptr(hm), _lib.ptr(fmin), _lib.ptr(indent),
                _lib.ptr(cam_u8) if cam_u8 is not None else 0, _lib.ptr(rows) if rows is not None else 0, B)
        _lib.check(rc, "tacex_taxim_defer_height_map_from_depth")
        # the armed pass holds raw device pointers until the render launches it: keep every buffer it reads or fills alive until then
        # (the depth may be a conversion copy the caller drops on return)
        self._deferred_keepalive = (depth, hm, fmin, indent, cam_u8, rows)

    # -- plugin interface -------------------------------------------------------------------------------
    def optical_simulation(self):
        """(num_envs, H, W, 3) float32 RGB in [0,1] (the reference docstring says 0..255, taxim_sim.py:83, wrongly)."""
        height_map, resized = self._tactile_height_map()
        have_min = (not resized) and self._frame_min_version == self.sensor._height_map_version
        W, H = self.cfg.tactile_img_res
        # marker-pixel outputs replace the full deformed-gel / mask frames while the fused tail is available
        compact = (self._keep_deformation and getattr(self, "_pix_z", None) is not None and not self.cfg.with_shadow
                   and self._taxim.fots_partials_per_env((H, W)) > 0)
        full = self._keep_deformation and not compact
        self._taxim.render_direct(
            height_map,
            with_shadow=self.cfg.with_shadow,
            press_depth=self._indentation_depth,
            orig_hm_fmt=False,
            out=self.tactile_rgb_img,
            frame_min=self._frame_min if have_min else None,
            frame_rows=self._frame_rows if (have_min and self._frame_rows_version == self.sensor._height_map_version) else None,
            z_out=self._deformed_gel if full else None,
            mask_out=self._contact_mask if full else None,
            obs_out=self.policy_obs,
        )
        self._deferred_keepalive = None  # the deferred depth pass (if any) has been enqueued; stream order protects its buffers from here on
        if self._keep_deformation:
            self._deformation_version = self.sensor._height_map_version if full else -1
            self._fots_compact_version = self.sensor._height_map_version if compact else -1
            n = self._taxim.fots_partials_per_env((H, W))  # 0 while the fused tail is disabled
            self._fots_partials_version = self.sensor._height_map_version if n == self._fots_partials.shape[1] and n > 0 else -1
        return self.tactile_rgb_img

    def compute_indentation_depth(self):
        """taxim_sim.py:115-131 in one reduction kernel; also caches the per-frame minimum."""
        if self._indent_version == self.sensor._height_map_version:
            return self._indentation_depth  # already filled by the fused depth -> height-map pass of the sensor
        height_map = self.sensor._data.output["height_map"]
        B, H, W = height_map.shape
        lib = _lib.load_library()
        hm = height_map if height_map.is_contiguous() else height_map.contiguous()
        with torch.cuda.device(hm.device):
            rc = lib.tacex_indentation_depth(
                _lib.ptr(hm), float(self.cfg.gelpad_height), float(self.cfg.gelpad_to_camera_min_distance),
                _lib.ptr(self._frame_min), _lib.ptr(self._indentation_depth), _lib.ptr(self._frame_rows), B, H, W,
                _lib.current_stream_handle(hm.device))
        _lib.check(rc, "tacex_indentation_depth")
        self._frame_min_version = self.sensor._height_map_version
        self._frame_rows_version = self.sensor._height_map_version
        self._indent_version = self.sensor._height_map_version
        return self._indentation_depth

    def reset(self):
        """taxim_sim.py:133-135: fresh zero indentation buffer, background into the simulator's image buffer.

        Both buffers may be SHARED with the sensor here (they are private copies in the reference): the sensor has
        already zeroed the indentation depth of exactly the envs being reset (GS:163) and its `tactile_rgb` output keeps
        the frame rendered just before this call (GS:182-183), so shared buffers are left alone - zeroing / overwriting
        them for all envs would wipe the state of envs that are still in contact."""
        if getattr(self.sensor, "_indentation_depth", None) is not self._indentation_depth:
            self._indentation_depth.zero_()
        out = getattr(self.sensor, "_data", None)
        out = out.output.get("tactile_rgb") if out is not None and out.output else None
        if out is None or out.data_ptr() != self.tactile_rgb_img.data_ptr():
            self.tactile_rgb_img[:] = self.background_img
        self._frame_min_version = -1
        self._indent_version = -1
        self._deformation_version = -1
        self._fots_partials_version = -1
        self._fots_compact_version = -1

    def _set_debug_vis_impl(self, debug_vis: bool):
        pass  # Kit UI windows of the reference (taxim_sim.py:137-213) are out of scope

    def _debug_vis_callback(self, event):
        pass
