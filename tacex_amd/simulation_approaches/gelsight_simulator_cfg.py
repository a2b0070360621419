from ..utils.configclass import configclass


@configclass
class GelSightSimulatorCfg:
    """Parent class of the simulation-approach cfgs (reference: gelsight_simulator_cfg.py:6-16)."""

    simulation_approach_class: type = None
    device: str = "cuda"
