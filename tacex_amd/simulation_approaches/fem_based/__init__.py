from .mani_skill_sim import ManiSkillSimulator
from .mani_skill_sim_cfg import ManiSkillSimulatorCfg

__all__ = ["ManiSkillSimulator", "ManiSkillSimulatorCfg"]
