"""Isaac-free restatement of IsaacLab's `SensorBase` / `SensorBaseCfg` bookkeeping.

`GelSightSensor` in the reference derives from `isaaclab.sensors.SensorBase`
(source/tacex/tacex/gelsight_sensor.py:31); IsaacLab itself is third-party and not in /root/reference,
so its public update/reset contract is restated here (SURVEY.md 8(b) "SensorBase semantics"):

  * per-env `_timestamp`, `_timestamp_last_update`, `_is_outdated`;
  * `update(dt, force_recompute)`: timestamp += dt; outdated if t - t_last + 1e-6 >= update_period;
    recompute now only if `force_recompute` (or history/visualisation), else lazily on `.data`;
  * `reset(env_ids)`: zero the timestamps and mark outdated.

Differences, all deliberate: no USD stage / timeline callbacks (initialisation is explicit or lazy, the
number of environments and the device come from the cfg), and the three bookkeeping vectors live on
the HOST so that `update()` never forces a device synchronisation (`nonzero()` on a device tensor
would) - they are identical for all envs in every reference task anyway.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence

import torch

from .utils.configclass import configclass


@configclass
class SensorBaseCfg:
    """Fields of IsaacLab's SensorBaseCfg that TacEx uses (+ num_envs, which Isaac derives from the stage)."""

    class_type: type = None
    prim_path: str = "/World/envs/env_.*/Sensor"
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False
    num_envs: int = 1
    """Number of environments (IsaacLab counts the prims matching `prim_path`; there is no stage here)."""


class SensorBase(ABC):
    def __init__(self, cfg: SensorBaseCfg):
        if cfg.history_length < 0:
            raise ValueError(f"History length must be greater than 0! Received: {cfg.history_length}")
        self.cfg = cfg
        self._is_initialized = False
        self._is_visualizing = False
        self._num_envs = int(cfg.num_envs)
        self._device = getattr(cfg, "device", "cuda")

    # -- properties ---------------------------------------------------------------------------------
    @property
    def is_initialized(self) -> bool:
        return self._is_initialized

    @property
    def num_instances(self) -> int:
        return self._num_envs

    @property
    def device(self) -> str:
        return self._device

    @property
    @abstractmethod
    def data(self):
        raise NotImplementedError

    # -- operations ---------------------------------------------------------------------------------
    def initialize(self):
        """Explicit counterpart of IsaacLab's timeline-PLAY callback."""
        if not self._is_initialized:
            self._initialize_impl()
            self._is_initialized = True

    def reset(self, env_ids: Sequence[int] | None = None):
        if env_ids is None:
            env_ids = slice(None)
        elif isinstance(env_ids, torch.Tensor):
            env_ids = env_ids.cpu()
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True

    def update(self, dt: float, force_recompute: bool = False):
        if not self._is_initialized:
            self.initialize()
        self._timestamp += dt
        self._is_outdated |= self._timestamp - self._timestamp_last_update + 1e-6 >= self.cfg.update_period
        if force_recompute or self._is_visualizing or (self.cfg.history_length > 0):
            self._update_outdated_buffers()

    # -- implementation hooks -------------------------------------------------------------------------
    def _initialize_impl(self):
        n = self._num_envs
        self._is_outdated = torch.ones(n, dtype=torch.bool)
        self._timestamp = torch.zeros(n, dtype=torch.float64)
        self._timestamp_last_update = torch.zeros(n, dtype=torch.float64)

    @abstractmethod
    def _update_buffers_impl(self, env_ids: Sequence[int]):
        raise NotImplementedError

    def _update_outdated_buffers(self):
        if not self._is_initialized:
            self.initialize()
        if bool(self._is_outdated.all()):
            outdated = slice(None)
            any_outdated = True
        else:
            outdated = self._is_outdated.nonzero().squeeze(-1)
            any_outdated = len(outdated) > 0
        if any_outdated:
            self._update_buffers_impl(outdated)
            self._timestamp_last_update[outdated] = self._timestamp[outdated]
            self._is_outdated[outdated] = False
