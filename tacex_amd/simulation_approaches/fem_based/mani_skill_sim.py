"""`ManiSkillSimulator` - FEM-based marker-motion plugin (ManiSkill-ViTac style on the UIPC gelpad), counterpart of
source/tacex/tacex/simulation_approaches/fem_based/mani_skill_sim.py:22-86.  The reference writes the flow of env 0
only (`self.marker_data[0] = marker_flow`, MS:84-85); here every env gets its own flow."""
from __future__ import annotations

from typing import TYPE_CHECKING

import torch

from ..gelsight_simulator import GelSightSimulator
from .sim import VisionTactileSensorUIPC

if TYPE_CHECKING:
    from ...gelsight_sensor import GelSightSensor
    from .mani_skill_sim_cfg import ManiSkillSimulatorCfg


class ManiSkillSimulator(GelSightSimulator):
    cfg: "ManiSkillSimulatorCfg"

    def __init__(self, sensor: "GelSightSensor", cfg: "ManiSkillSimulatorCfg"):
        self.sensor = sensor
        self.camera = None
        self.gelpad_uipc = self.sensor.gelpad_obj  # UipcObject of the gelpad (GS:34, MS:37)
        super().__init__(sensor=sensor, cfg=cfg)

    def _initialize_impl(self):
        self._device = self.sensor.device if self.cfg.device is None else self.cfg.device
        self._num_envs = self.sensor._num_envs
        self._indentation_depth = torch.zeros((self._num_envs,), device=self.sensor._device)
        if self.gelpad_uipc is None or getattr(self.gelpad_uipc, "_uipc_sim", None) is None:
            raise RuntimeError("ManiSkillSimulator needs GelSightSensor(cfg, gelpad_obj=<UipcObject attached to a UipcSim>)")
        sim = self.gelpad_uipc._uipc_sim
        if sim.num_envs != self._num_envs:
            raise RuntimeError(f"UipcSim has {sim.num_envs} envs, the sensor {self._num_envs}")
        self.marker_motion_sim = VisionTactileSensorUIPC(
            self.gelpad_uipc, sim,
            cam_pos_w=torch.as_tensor(self.cfg.camera_pos_w, dtype=torch.float64),
            cam_quat_w_ros=torch.as_tensor(self.cfg.camera_quat_w_ros, dtype=torch.float64),
            tactile_img_width=self.cfg.tactile_img_res[0], tactile_img_height=self.cfg.tactile_img_res[1],
            marker_interval_range=self.cfg.marker_interval_range, marker_rotation_range=self.cfg.marker_rotation_range,
            marker_translation_range=self.cfg.marker_translation_range, marker_pos_shift_range=self.cfg.marker_pos_shift_range,
            marker_random_noise=self.cfg.marker_random_noise,
            marker_lose_tracking_probability=self.cfg.marker_lose_tracking_probability, normalize=self.cfg.normalize,
            num_markers=self.cfg.marker_params.num_markers, camera_params=self.cfg.camera_params)
        self.marker_data = torch.zeros((self._num_envs, 2, self.cfg.marker_params.num_markers, 2), device=self._device)

    def marker_motion_simulation(self):
        # static marker grid (the shipped cfgs): one launch from the FEM state straight into marker_data; otherwise the general path
        if self.marker_data.dtype == torch.float32 and self.marker_motion_sim.gen_marker_flow_fused(out_f32=self.marker_data) is not None:
            return self.marker_data
        self.marker_data[:] = self.marker_motion_sim.gen_marker_flow().to(self.marker_data.dtype)
        return self.marker_data

    def reset(self):
        self._indentation_depth = torch.zeros((self._num_envs,), device=self._device)

    def _set_debug_vis_impl(self, debug_vis: bool):
        pass

    def _debug_vis_callback(self, event):
        pass
