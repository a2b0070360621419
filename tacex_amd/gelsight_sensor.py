"""`GelSightSensor` - the sensor boundary, drop-in for source/tacex/tacex/gelsight_sensor.py:31-378.

Same constructor, properties (`data`, `frame`, `tactile_image_shape`, `camera_resolution`,
`indentation_depth`), `update()` / `reset()` semantics and persistent output buffers that are written in
place.  The Isaac `TiledCamera` is replaced by an injected depth image (cfg.sensor_camera_cfg.depth_source
or `set_camera_depth`); everything downstream of that image runs in HIP kernels.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import TYPE_CHECKING

import os

import torch

from . import _lib
from .gelsight_sensor_data import GelSightSensorData
from .sensor_base import SensorBase as _LocalSensorBase
from .simulation_approaches.gelsight_simulator import GelSightSimulator

# Under a real IsaacLab the sensor derives from `isaaclab.sensors.SensorBase` like the reference's (gelsight_sensor.py:31): the scene then
# drives it through IsaacLab's own timeline / update / reset machinery.  Without IsaacLab (this repo's tests, the bench, any Isaac-free host)
# the restatement in sensor_base.py stands in.  TACEX_SENSOR_BASE=local forces the restatement (untested against IsaacLab here: it is not
# installable in the build container - INTEGRATION.md section 2).
SensorBase = _LocalSensorBase
if os.environ.get("TACEX_SENSOR_BASE", "auto") != "local":
    try:
        from isaaclab.sensors import SensorBase as _IsaacSensorBase  # type: ignore

        SensorBase = _IsaacSensorBase
    except Exception:  # ImportError, or Kit not running
        pass

_DEFER_DEPTH = os.environ.get("TACEX_DEFER_DEPTH", "1") != "0"  # A/B switch of _can_defer_depth_pass

if TYPE_CHECKING:
    from .gelsight_sensor_cfg import GelSightSensorCfg


class GelSightSensor(SensorBase):
    cfg: "GelSightSensorCfg"

    def __init__(self, cfg: "GelSightSensorCfg", gelpad_obj=None):
        self.cfg = cfg
        self._prim_view = None
        self.camera = None  # there is no TiledCamera here; see set_camera_depth
        self.gelpad_obj = gelpad_obj
        self._indentation_depth: torch.Tensor = None
        self.optical_simulator: GelSightSimulator = None
        self.marker_motion_simulator: GelSightSimulator = None
        self.compute_indentation_depth_func = None
        self._data = GelSightSensorData()
        self._data.output = dict.fromkeys(self.cfg.data_types, None)
        self._is_spawned = False
        self._camera_depth_m = None
        self._height_map_version = 0
        self._group = None  # set by GelSightSensorGroup: the member's buffers are views of the group's core sensor
        # SensorBase sets _num_envs / _device, which the simulator constructors read
        super().__init__(self.cfg)

        # instantiate the simulation approaches named in the cfg (gelsight_sensor.py:59-77)
        if self.cfg.optical_sim_cfg is not None:
            self.optical_simulator = self.cfg.optical_sim_cfg.simulation_approach_class(
                sensor=self, cfg=self.cfg.optical_sim_cfg
            )
        if self.cfg.marker_motion_sim_cfg is not None:
            if (self.optical_simulator is not None) and (
                self.cfg.optical_sim_cfg.simulation_approach_class
                == self.cfg.marker_motion_sim_cfg.simulation_approach_class
            ):
                self.marker_motion_simulator = self.optical_simulator
            else:
                self.marker_motion_simulator = self.cfg.marker_motion_sim_cfg.simulation_approach_class(
                    sensor=self, cfg=self.cfg.marker_motion_sim_cfg
                )

    # -- properties (gelsight_sensor.py:107-140) --------------------------------------------------------
    @property
    def data(self) -> GelSightSensorData:
        self._update_outdated_buffers()
        return self._data

    @property
    def frame(self) -> torch.Tensor:
        """Per-env update counter (gelsight_sensor.py:345).  Full-batch updates only bump a host integer (no kernel per
        step); the tensor is materialised on read."""
        if self._frame_pending:
            self._frame += self._frame_pending
            self._frame_pending = 0
        return self._frame

    @property
    def tactile_image_shape(self) -> tuple[int, int, int]:
        return self.cfg.optical_sim_cfg.tactile_img_res[1], self.cfg.optical_sim_cfg.tactile_img_res[0], 3

    @property
    def camera_resolution(self) -> tuple[int, int]:
        return self.cfg.sensor_camera_cfg.resolution[0], self.cfg.sensor_camera_cfg.resolution[1]

    @property
    def indentation_depth(self):
        """How deep objects are inside the gel pad [mm]."""
        return self._indentation_depth

    @property
    def prim_view(self):
        return self._prim_view

    # -- depth injection (replaces TiledCamera, gelsight_sensor.py:229-263) ---------------------------------
    def set_camera_depth(self, depth_m: torch.Tensor):
        """Provide the camera depth image in metres: (num_envs, Hc, Wc) or (num_envs, Hc, Wc, 1), float32."""
        if self._group is not None:
            return self._group._set_member_depth(self, depth_m)
        if depth_m.dim() == 4:
            depth_m = depth_m[..., 0]
        W, H = self.camera_resolution
        if tuple(depth_m.shape) != (self._num_envs, H, W):
            raise RuntimeError(
                f"camera depth has shape {tuple(depth_m.shape)}, expected ({self._num_envs}, {H}, {W}) "
                "(num_envs, camera height, camera width)"
            )
        self._camera_depth_m = depth_m

    def set_height_map(self, height_map_mm: torch.Tensor):
        """Provide the height map (mm, (num_envs, Hc, Wc)) directly instead of a camera depth image: copied into the
        persistent `output["height_map"]` buffer; cached per-frame minima / indentation depths are invalidated."""
        hm = self._data.output["height_map"]
        if tuple(height_map_mm.shape) != tuple(hm.shape):
            raise RuntimeError(f"height map has shape {tuple(height_map_mm.shape)}, expected {tuple(hm.shape)}")
        if height_map_mm.data_ptr() != hm.data_ptr():
            hm.copy_(height_map_mm)
        self.mark_height_map_dirty()

    def mark_height_map_dirty(self):
        """Call after writing `output["height_map"]` in place."""
        self._height_map_version += 1
        if self._group is not None:
            self._group.core.mark_height_map_dirty()

    def set_height_map_source(self, source):
        """Fill the height map from an on-device source (e.g. `IndenterHeightMapSource`) instead of a camera depth image
        (SURVEY 8f n1).  `source.fill(hm, frame_min, indent, gelpad_height, gelpad_to_camera_min_distance)`."""
        if self._group is not None:
            raise RuntimeError("a grouped sensor takes its height map from the camera depth or set_height_map (one source per group)")
        self._height_map_source = source

    def _read_camera_depth(self):
        src = self.cfg.sensor_camera_cfg.depth_source if self.cfg.sensor_camera_cfg is not None else None
        if src is not None:
            self.set_camera_depth(src())
        return self._camera_depth_m

    def initialize(self):
        """Explicit initialisation (IsaacLab does it from its timeline-PLAY callback; both end in `_initialize_impl`)."""
        if hasattr(super(), "initialize"):
            return super().initialize()
        if not self._is_initialized:
            self._initialize_impl()
            self._is_initialized = True

    # -- reset (gelsight_sensor.py:147-197) -----------------------------------------------------------------
    def reset(self, env_ids: Sequence[int] | None = None):
        if self._group is not None:
            return self._group._member_reset(self, env_ids)
        if not self._is_initialized:
            # lazy initialisation brings the sensor's own buffers to their reset state; the reset the caller asked for (its `env_ids`, and
            # the FEM pads of exactly those envs) then runs like any other
            self.initialize()
        self._reset_impl(env_ids)

    def _reset_impl(self, env_ids, reset_gelpad: bool = True):
        super().reset(env_ids)
        # FEM gelpad (tacex_uipc): the pad of a reset env goes back to its rest shape with the sensor.  In the reference the scene resets
        # its assets one by one and `UipcObject.reset` is a TODO stub (uipc_object.py:280-286) - its UIPC scenes hold one env; here the
        # sensor owns the reference to the pad, so a task that resets `env_ids` of the sensor gets a consistent pad / image pair.
        # `cfg.reset_gelpad_with_sensor = False` leaves the pad to the caller.  Only an explicit `reset()` touches the pad: initialising a
        # sensor (`reset_gelpad=False`) must not wipe a pad that was placed with `write_vertex_positions_to_sim` or has already stepped.
        pad_sim = getattr(self.gelpad_obj, "_uipc_sim", None)
        if reset_gelpad and pad_sim is not None and getattr(pad_sim, "_handle", None) is not None \
                and getattr(self.cfg, "reset_gelpad_with_sensor", True):
            self.gelpad_obj.reset(env_ids)
        if env_ids is None:
            env_ids = self._ALL_INDICES
        self._indentation_depth[env_ids] = 0
        self._data.output["height_map"][env_ids] = 0
        self._height_map_version += 1
        if "camera_depth" in self._data.output and self._data.output["camera_depth"] is not None:
            self._data.output["camera_depth"][env_ids] = 0
        # simulate optical/marker output without indentation, then reset the simulators (reference order,
        # gelsight_sensor.py:182-193: the render happens BEFORE optical_simulator.reset())
        # Per-env state of envs that are NOT being reset survives, as in the reference: its simulators' reset() only
        # re-allocate their private indentation buffers (TS:133-135, FS:206-208); the trajectory of an env is cleared by
        # the marker simulation itself when its indentation depth is 0 (FS:176-177), which holds for `env_ids` here.
        if (self.optical_simulator is not None) and ("tactile_rgb" in self._data.output):
            res = self.optical_simulator.optical_simulation()
            if res.data_ptr() != self._data.output["tactile_rgb"].data_ptr():
                self._data.output["tactile_rgb"][:] = res
            self.optical_simulator.reset()
        if (self.marker_motion_simulator is not None) and ("marker_motion" in self._data.output):
            res = self.marker_motion_simulator.marker_motion_simulation()
            if res.data_ptr() != self._data.output["marker_motion"].data_ptr():
                self._data.output["marker_motion"][:] = res
            self._data.output["init_marker_pos"] = ([0], [0])
            self.marker_motion_simulator.reset()
        self._frame[env_ids] = -self._frame_pending  # frame == 0 once the pending full-batch increments are folded in

    # -- initialisation (gelsight_sensor.py:203-337) -----------------------------------------------------------
    def _initialize_impl(self):
        super()._initialize_impl()
        if self.cfg.device is not None:
            self._device = self.cfg.device
        dev = torch.device(self._device)
        if dev.type != "cuda":
            raise _lib.TacexHipError(f"GelSightSensor needs an AMD GPU device, got '{self._device}' (no CPU fallback)")
        _lib.require_gpu(dev.index or 0)
        self._ALL_INDICES = torch.arange(self._num_envs, device=self._device, dtype=torch.long)
        self._frame = torch.zeros(self._num_envs, device=self._device, dtype=torch.long)
        self._frame_pending = 0
        self._indentation_depth = torch.zeros((self._num_envs,), device=self._device)
        Wc, Hc = self.camera_resolution
        self._data.output["height_map"] = torch.zeros((self._num_envs, Hc, Wc), device=self._device)

        if self.optical_simulator is not None:
            self.optical_simulator._initialize_impl()
        if self.marker_motion_simulator is not None and self.marker_motion_simulator is not self.optical_simulator:
            self.marker_motion_simulator._initialize_impl()
        elif self.marker_motion_simulator is not None and not hasattr(self.marker_motion_simulator, "marker_data"):
            self.marker_motion_simulator._initialize_impl()

        if "camera_depth" in self.cfg.data_types:
            self._data.output["camera_depth"] = torch.zeros((self._num_envs, Hc, Wc, 1), dtype=torch.uint8, device=self._device)
        if "camera_rgb" in self.cfg.data_types:
            self._data.output["camera_rgb"] = torch.zeros((self._num_envs, Hc, Wc, 3), device=self._device)
        if "tactile_rgb" in self.cfg.data_types:
            if self.cfg.optical_sim_cfg is None:
                raise RuntimeError("data type 'tactile_rgb' needs an optical_sim_cfg")
            W, H = self.cfg.optical_sim_cfg.tactile_img_res
            buf = getattr(self.optical_simulator, "tactile_rgb_img", None)
            if buf is not None and tuple(buf.shape) == (self._num_envs, H, W, 3) and buf.is_cuda:
                # the simulator renders straight into the sensor's persistent output buffer: the reference's
                # per-step `output["tactile_rgb"][:] = ...` (GS:375) is a 236 MB device copy at 256 envs
                self._data.output["tactile_rgb"] = buf
            else:
                self._data.output["tactile_rgb"] = torch.zeros((self._num_envs, H, W, 3), device=self._device)
            if getattr(self.optical_simulator, "policy_obs", None) is not None:
                self._data.output["tactile_rgb_obs"] = self.optical_simulator.policy_obs
        if "marker_motion" in self.cfg.data_types:
            if self.cfg.marker_motion_sim_cfg is None:
                raise RuntimeError("data type 'marker_motion' needs a marker_motion_sim_cfg")
            nm = self.cfg.marker_motion_sim_cfg.marker_params.num_markers
            buf = getattr(self.marker_motion_simulator, "marker_data", None)
            if buf is not None and tuple(buf.shape) == (self._num_envs, 2, nm, 2):
                self._data.output["marker_motion"] = buf
            else:
                self._data.output["marker_motion"] = torch.zeros((self._num_envs, 2, nm, 2), device=self._device)

        # how the indentation depth is computed (gelsight_sensor.py:321-329)
        if self.cfg.compute_indentation_depth_class == "optical_sim" and self.optical_simulator is not None:
            self.compute_indentation_depth_func = self.optical_simulator.compute_indentation_depth
        elif self.cfg.compute_indentation_depth_class == "marker_motion_sim" and self.marker_motion_simulator is not None:
            self.compute_indentation_depth_func = self.marker_motion_simulator.compute_indentation_depth
        else:
            self.compute_indentation_depth_func = None
        # the provider fills its own (B,) buffer in the fused depth pass: share it instead of copying it every step
        provider = getattr(self.compute_indentation_depth_func, "__self__", None)
        buf = getattr(provider, "_indentation_depth", None)
        if isinstance(buf, torch.Tensor) and buf.is_cuda and tuple(buf.shape) == (self._num_envs,):
            self._indentation_depth = buf
        self._is_initialized = True
        self._reset_impl(None, reset_gelpad=False)

    # -- per-step update (gelsight_sensor.py:342-378) ------------------------------------------------------------
    def _update_buffers_impl(self, env_ids: Sequence[int]):
        if self._group is not None:  # one evaluation for all sensors of the group (gelsight_sensor_group.py)
            return self._group._member_update(self, env_ids)
        # like the reference, env_ids only selects which frame counters advance: all envs are recomputed
        if isinstance(env_ids, slice):
            self._frame_pending += 1
        else:
            self._frame[env_ids.to(self._frame.device)] += 1

        if self.compute_indentation_depth_func is not None:
            self._get_height_map(defer=self._can_defer_depth_pass())
            res = self.compute_indentation_depth_func()
            if res.data_ptr() != self._indentation_depth.data_ptr():
                self._indentation_depth[:] = res

        if "camera_depth" in self._data.output:
            self._get_camera_depth()

        if (self.optical_simulator is not None) and ("tactile_rgb" in self.cfg.data_types):
            res = self.optical_simulator.optical_simulation()
            if res.data_ptr() != self._data.output["tactile_rgb"].data_ptr():
                self._data.output["tactile_rgb"][:] = res

        if (self.marker_motion_simulator is not None) and ("marker_motion" in self.cfg.data_types):
            res = self.marker_motion_simulator.marker_motion_simulation()
            if res.data_ptr() != self._data.output["marker_motion"].data_ptr():
                self._data.output["marker_motion"][:] = res

    # -- camera -> height map (gelsight_sensor.py:557-593) ---------------------------------------------------------
    def _fused_targets(self):
        """Buffers of the indentation-depth provider the fused depth kernel may fill directly."""
        sim = None
        if self.compute_indentation_depth_func is not None:
            sim = getattr(self.compute_indentation_depth_func, "__self__", None)
        if sim is not None and hasattr(sim, "_frame_min") and hasattr(sim.cfg, "gelpad_height"):
            return sim
        return None

    def _can_defer_depth_pass(self) -> bool:
        """True when the optical simulator renders in this update from exactly the buffers the depth pass fills: the pass is then handed
        to the render (`tacex_taxim_defer_height_map_from_depth`), which runs it chunk by chunk beside the band levels of the previous
        chunk instead of as one launch ahead of them.  `TACEX_DEFER_DEPTH=0` keeps the two launches apart (A/B)."""
        if _DEFER_DEPTH is False:
            return False
        sim = self.optical_simulator
        if sim is None or "tactile_rgb" not in self.cfg.data_types or not hasattr(sim, "defer_height_map_from_depth"):
            return False
        if sim is not self._fused_targets() or sim._indentation_depth.data_ptr() != self._indentation_depth.data_ptr():
            return False
        W, H = sim.cfg.tactile_img_res
        return tuple(self._data.output["height_map"].shape[1:]) == (H, W)

    def _get_height_map(self, defer: bool = False):
        src = getattr(self, "_height_map_source", None)
        if src is not None:  # analytic source: height map, frame minimum and indentation depth in one launch
            hm = self._data.output["height_map"]
            sim = self._fused_targets()
            fmin = sim._frame_min if sim is not None else self._scratch_min()
            src.fill(hm, fmin, sim._indentation_depth if sim is not None else None,
                     float(sim.cfg.gelpad_height) if sim is not None else 0.0,
                     float(sim.cfg.gelpad_to_camera_min_distance) if sim is not None else 0.0)
            self._height_map_version += 1
            if sim is not None:
                sim._frame_min_version = self._height_map_version
                sim._indent_version = self._height_map_version
            return hm
        depth = self._read_camera_depth()
        if depth is None:
            # no camera: the caller writes output["height_map"] itself (set_height_map, or in place).  An in-place write
            # cannot be seen from here, so every update treats the map as new: the per-frame minimum and the indentation
            # depth are recomputed (one small reduction kernel) instead of being served from the previous map's cache.
            self._height_map_version += 1
            return self._data.output["height_map"]
        hm = self._data.output["height_map"]
        near, far = self.cfg.sensor_camera_cfg.clipping_range
        sim = self._fused_targets()
        want_u8 = "camera_depth" in self._data.output and self._data.output["camera_depth"] is not None
        depth = depth if (depth.is_contiguous() and depth.dtype == torch.float32) else depth.float().contiguous()
        lib = _lib.load_library()
        B, H, W = hm.shape
        fmin = sim._frame_min if sim is not None else self._scratch_min()
        indent = sim._indentation_depth if sim is not None else None
        rows = getattr(sim, "_frame_rows", None) if indent is not None else None  # contact row range per frame (band skipping)
        if defer and sim is not None and indent is not None:
            sim.defer_height_map_from_depth(depth, near, far, hm, fmin, indent, self._data.output["camera_depth"] if want_u8 else None, rows)
        else:
            with torch.cuda.device(hm.device):
                rc = lib.tacex_height_map_from_depth(
                    _lib.ptr(depth), float(near), float(far),
                    float(sim.cfg.gelpad_height) if sim is not None else 0.0,
                    float(sim.cfg.gelpad_to_camera_min_distance) if sim is not None else 0.0,
                    _lib.ptr(hm), _lib.ptr(fmin), _lib.ptr(indent),
                    _lib.ptr(self._data.output["camera_depth"]) if want_u8 else 0,
                    _lib.ptr(rows) if rows is not None else 0,
                    B, H, W, _lib.current_stream_handle(hm.device))
            _lib.check(rc, "tacex_height_map_from_depth")
        self._height_map_version += 1
        if sim is not None:
            sim._frame_min_version = self._height_map_version
            sim._indent_version = self._height_map_version
            if rows is not None:
                sim._frame_rows_version = self._height_map_version
        self._camera_depth_version = self._height_map_version
        return hm

    def _scratch_min(self):
        if getattr(self, "_fmin_scratch", None) is None:
            self._fmin_scratch = torch.zeros((self._num_envs,), device=self._device)
        return self._fmin_scratch

    def _get_camera_depth(self):
        # the uint8 image is produced by the same pass as the height map; only redo it if that did not run
        if getattr(self, "_camera_depth_version", -1) != self._height_map_version and self._read_camera_depth() is not None:
            self._get_height_map()
        return self._data.output["camera_depth"]

    # -- debug vis: Kit UI of the reference (gelsight_sensor.py:380-555) is out of scope -----------------------------
    def _set_debug_vis_impl(self, debug_vis: bool):
        pass

    def _debug_vis_callback(self, event):
        pass
