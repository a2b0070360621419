from .fots_marker_sim import FOTSMarkerSimulator
from .fots_marker_sim_cfg import FOTSMarkerSimulatorCfg

__all__ = ["FOTSMarkerSimulator", "FOTSMarkerSimulatorCfg"]
