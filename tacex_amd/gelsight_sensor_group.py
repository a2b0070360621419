"""`GelSightSensorGroup` - several GelSightSensors of ONE configuration evaluated as one launch sequence.

The reference's two-finger scenes build one `GelSightSensor` per finger (`gsmini_left` / `gsmini_right`,
source/tacex_tasks/.../factory_env_cfg.py:192-213, factory_env.py:190-194): two sensors, two sets of TaximTorch tables, two
render calls per step.  Every op of the tactile path is per frame, so the frames of n sensors over B envs ARE one batch of n*B
frames ("the build may batch both sensors into one 2B-frame launch").  A group keeps ONE core sensor of
n*B frames; the member sensors stay the objects a task holds - their `data.output[...]`, `indentation_depth`, `frame` are views
of the core's buffers, `update()` / `reset()` / lazy `.data` keep the SensorBase contract per member - but the kernels run once
per step over all of them: half the launches and launch drains, and twice the items for the streaming tail's heaviest-first
order to balance.

Contract: the depth inputs of ALL members must be in place (set_camera_depth / depth_source / camera_depth_buffer) before the
FIRST member of a step is updated - the order IsaacLab's scene update gives (render, then sensors).  A member that is updated a
second time before the others were asked starts the next evaluation.
"""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .gelsight_sensor import GelSightSensor
from .sensor_base import SensorBase


def _cfg_signature(cfg):
    """What must agree between members: everything the kernels and tables depend on (not the prim path / env count)."""
    d = cfg.to_dict() if hasattr(cfg, "to_dict") else dict(vars(cfg))
    for k in ("prim_path", "class_type"):
        d.pop(k, None)
    cam = d.get("sensor_camera_cfg")
    if isinstance(cam, dict):
        cam.pop("depth_source", None)
        cam.pop("prim_path_appendix", None)
    mm = d.get("marker_motion_sim_cfg")
    if isinstance(mm, dict):
        mm.pop("yaw_source", None)
    return repr(d)


class _MemberSimulator:
    """A member's view of the core's simulator: per-env tensors are slices, everything else passes through."""

    _SLICED = ("tactile_rgb_img", "policy_obs", "marker_data", "_indentation_depth", "_frame_min", "_frame_rows", "_traj_state",
               "_pix_z", "_pix_m", "_deformed_gel", "_contact_mask")

    def __init__(self, core_sim, sl: slice):
        object.__setattr__(self, "_core", core_sim)
        object.__setattr__(self, "_sl", sl)

    def __getattr__(self, name):
        v = getattr(self._core, name)
        if name in self._SLICED and isinstance(v, torch.Tensor):
            return v[self._sl]
        return v

    @property
    def theta(self):
        return self._core.theta[self._sl]

    def indenter_yaw_buffer(self) -> torch.Tensor:
        """(B,) float32 view of this member's rows of the core's yaw vector: a producer may write the yaw straight into it (no copy in
        set_indenter_yaw then - the counterpart of GelSightSensorGroup.camera_depth_buffer)."""
        core = self._core
        if core.theta.shape[0] != core._num_envs or not core.theta.is_contiguous():
            core.theta = torch.zeros((core._num_envs,), device=core._device)
        return core.theta[self._sl]

    def set_indenter_yaw(self, theta: torch.Tensor):
        """Yaw of this member's indenters: written into the member's rows of the core's (n*B,) yaw vector."""
        dst = self.indenter_yaw_buffer()
        theta = theta.reshape(-1)
        if not (theta.data_ptr() == dst.data_ptr() and theta.dtype == dst.dtype and theta.shape == dst.shape):  # else: already in place
            dst.copy_(theta)


class GelSightSensorGroup:
    def __init__(self, sensors: Sequence[GelSightSensor]):
        sensors = list(sensors)
        if len(sensors) < 1:
            raise ValueError("a sensor group needs at least one sensor")
        sig = _cfg_signature(sensors[0].cfg)
        B = sensors[0]._num_envs
        for s in sensors:
            if not isinstance(s, GelSightSensor):
                raise TypeError(f"not a GelSightSensor: {type(s)}")
            if s._is_initialized or getattr(s, "_group", None) is not None:
                raise RuntimeError("group sensors before they are initialised (the group owns their buffers)")
            if s.gelpad_obj is not None:
                raise RuntimeError("sensors with a FEM gelpad object cannot be grouped (their marker flow reads per-sensor meshes)")
            if s._num_envs != B or _cfg_signature(s.cfg) != sig:
                raise RuntimeError("sensors of a group must share one configuration (resolution, calibration, simulators, env count)")
        self.sensors = sensors
        self.num_envs = B
        cfg = sensors[0].cfg.copy()
        cfg.num_envs = B * len(sensors)
        cfg.sensor_camera_cfg.depth_source = None  # members feed their slices
        if cfg.marker_motion_sim_cfg is not None and hasattr(cfg.marker_motion_sim_cfg, "yaw_source"):
            cfg.marker_motion_sim_cfg.yaw_source = None
        self.core = GelSightSensor(cfg)
        self.core.initialize()
        W, H = self.core.camera_resolution
        self._depth = None  # (n*B, Hc, Wc) camera depth of all members, allocated when a member provides one
        self._served: set[int] = set()
        self._inputs_dirty = False  # a member's depth / yaw was handed over since the last evaluation
        for i, s in enumerate(sensors):
            self._adopt(s, i)

    # -- member wiring --------------------------------------------------------------------------------------------------------------
    def _adopt(self, s: GelSightSensor, i: int):
        B = self.num_envs
        sl = slice(i * B, (i + 1) * B)
        core = self.core
        s._group, s._group_index, s._group_slice = self, i, sl
        SensorBase._initialize_impl(s)
        s._device = core._device
        s._ALL_INDICES = torch.arange(B, device=core._device, dtype=torch.long)
        s._frame = torch.zeros(B, device=core._device, dtype=torch.long)
        s._frame_pending = 0
        s._indentation_depth = core._indentation_depth[sl]
        for k, v in core._data.output.items():
            s._data.output[k] = v[sl] if (isinstance(v, torch.Tensor) and v.shape[:1] == (B * len(self.sensors),)) else v
        if core.optical_simulator is not None:
            s.optical_simulator = _MemberSimulator(core.optical_simulator, sl)
        if core.marker_motion_simulator is not None:
            s.marker_motion_simulator = (s.optical_simulator if core.marker_motion_simulator is core.optical_simulator
                                         else _MemberSimulator(core.marker_motion_simulator, sl))
        s.compute_indentation_depth_func = core.compute_indentation_depth_func
        s._is_initialized = True

    def camera_depth_buffer(self, s: GelSightSensor) -> torch.Tensor:
        """(B, Hc, Wc) float32 view a depth producer may write straight into (no copy in set_camera_depth then)."""
        if self._depth is None:
            W, H = self.core.camera_resolution
            self._depth = torch.zeros((self.core._num_envs, H, W), device=self.core._device)
            self.core.set_camera_depth(self._depth)
        return self._depth[s._group_slice]

    def _set_member_depth(self, s: GelSightSensor, depth_m: torch.Tensor):
        if depth_m.dim() == 4:
            depth_m = depth_m[..., 0]
        buf = self.camera_depth_buffer(s)
        if tuple(depth_m.shape) != tuple(buf.shape):
            raise RuntimeError(f"camera depth has shape {tuple(depth_m.shape)}, expected {tuple(buf.shape)} (num_envs, camera height, camera width)")
        if depth_m.data_ptr() != buf.data_ptr():
            buf.copy_(depth_m)
        # (a producer that writes into camera_depth_buffer() in place hands nothing over: like a plain sensor's in-place depth, it is
        #  read by whichever update evaluates next)
        self._inputs_dirty = True

    # -- evaluation -------------------------------------------------------------------------------------------------------------------
    def _member_update(self, s: GelSightSensor, env_ids):
        if isinstance(env_ids, slice):
            s._frame_pending += 1
        else:
            s._frame[env_ids.to(s._frame.device)] += 1
        i = s._group_index
        # ONE evaluation per step when every member's inputs are in place before the first member updates (depth sources / yaw sources
        # are read here; a task that calls set_camera_depth() hands all depths over first).  A depth handed over AFTER the step's
        # evaluation - member 0 set and updated, then member 1 set and updated - makes the next update evaluate again: joining would
        # serve that member the frames rendered from its PREVIOUS depth.
        if self._served and i not in self._served and not self._inputs_dirty:
            self._served.add(i)  # this step's evaluation already covered the member
            return
        for m in self.sensors:  # every member's depth source is read before the one evaluation of the step
            src = m.cfg.sensor_camera_cfg.depth_source if m.cfg.sensor_camera_cfg is not None else None
            if src is not None:
                self._set_member_depth(m, src())
            mm = m.cfg.marker_motion_sim_cfg
            if mm is not None and getattr(mm, "yaw_source", None) is not None:
                m.marker_motion_simulator.set_indenter_yaw(mm.yaw_source())
        self.core._update_buffers_impl(slice(None))
        self._served = {i}
        self._inputs_dirty = False

    def _member_reset(self, s: GelSightSensor, env_ids):
        SensorBase.reset(s, env_ids)
        if env_ids is None:
            ids = s._ALL_INDICES
        else:
            ids = torch.as_tensor(env_ids, device=self.core._device, dtype=torch.long)
        self.core._reset_impl(ids + s._group_index * self.num_envs)
        s._frame[ids] = -s._frame_pending
        # the core's outputs now hold the reset render of these envs: whichever member asks next must trigger a fresh evaluation (leaving
        # the other members in `_served` would let the reset member "join" the evaluation that ran BEFORE its reset)
        self._served.clear()

    def update(self, dt: float, force_recompute: bool = False):
        """Convenience: update every member (ONE evaluation)."""
        for s in self.sensors:
            s.update(dt, force_recompute=force_recompute)

    def reset(self, env_ids: Sequence[int] | None = None):
        for s in self.sensors:
            s.reset(env_ids)
