from .gelsight_simulator import GelSightSimulator
from .gelsight_simulator_cfg import GelSightSimulatorCfg

__all__ = ["GelSightSimulator", "GelSightSimulatorCfg"]
