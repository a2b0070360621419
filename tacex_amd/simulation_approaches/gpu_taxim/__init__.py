from .taxim_sim import TaximSimulator
from .taxim_sim_cfg import TaximSimulatorCfg

__all__ = ["TaximSimulator", "TaximSimulatorCfg"]
