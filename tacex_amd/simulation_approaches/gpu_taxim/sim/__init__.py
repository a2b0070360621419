from ....calibration import CALIB_GELSIGHT, CALIB_GELSIGHT_MINI
from .taxim import Taxim
from .taxim_hip import TaximHip

__all__ = ["CALIB_GELSIGHT", "CALIB_GELSIGHT_MINI", "Taxim", "TaximHip"]
