"""Plugin ABC for GelSight simulation approaches - same interface as the reference's
source/tacex/tacex/simulation_approaches/gelsight_simulator.py:17-79."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import TYPE_CHECKING

if TYPE_CHECKING:
    from ..gelsight_sensor import GelSightSensor
    from .gelsight_simulator_cfg import GelSightSimulatorCfg


class GelSightSimulator(ABC):
    """Base class for implementing an optical / marker simulation approach."""

    def __init__(self, sensor: "GelSightSensor", cfg: "GelSightSimulatorCfg"):
        self.cfg = cfg
        self.sensor = sensor
        # use the same device as the sensor unless the cfg names one (gelsight_simulator.py:24-28)
        self._device = self.sensor.device if self.cfg.device is None else self.cfg.device

    @abstractmethod
    def _initialize_impl(self):
        raise NotImplementedError

    def optical_simulation(self):
        """Simulates the optical output of a tactile sensor."""
        raise NotImplementedError

    def marker_motion_simulation(self):
        """Simulates the marker motion of a tactile sensor."""
        raise NotImplementedError

    def compute_indentation_depth(self):
        """Computes how deep the indenter is pressed into the gelpad."""
        raise NotImplementedError

    @abstractmethod
    def reset(self):
        raise NotImplementedError

    def _set_debug_vis_impl(self, debug_vis: bool):
        raise NotImplementedError(f"Debug visualization is not implemented for {self.__class__.__name__}.")

    def _debug_vis_callback(self, event):
        raise NotImplementedError(f"Debug visualization is not implemented for {self.__class__.__name__}.")
