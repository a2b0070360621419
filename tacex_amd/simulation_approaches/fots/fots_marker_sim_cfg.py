from typing import Callable

from ...utils.configclass import configclass
from ..gelsight_simulator_cfg import GelSightSimulatorCfg
from .fots_marker_sim import FOTSMarkerSimulator


@configclass
class FOTSMarkerSimulatorCfg(GelSightSimulatorCfg):
    """Same fields as the reference's fots/fots_marker_sim_cfg.py:14-75; `frame_transformer_cfg` (an IsaacLab
    FrameTransformerCfg) is replaced by `yaw_source`."""

    simulation_approach_class: type = FOTSMarkerSimulator
    calib_folder_path: str = ""
    device: str = None
    with_shadow: bool = False
    tactile_img_res: tuple = (240, 320)  # (sic) the reference default is (W,H)-swapped; presets override to (320, 240)
    lamb: list = []
    """Unused, as in the reference: the exponents are hard-coded (fots_marker_sim.py:77)."""
    ball_radius: float = 4.70 / 2
    mm_to_pixel: float = 19.58
    pyramid_kernel_size: list = []
    kernel_size: int = 0

    @configclass
    class MarkerParams:
        num_markers_col: int = 11
        num_markers_row: int = 9
        num_markers: int = 99
        x0: float = 15.0
        y0: float = 26.0
        dx: float = 26.0
        dy: float = 29.0

    marker_params: MarkerParams = MarkerParams()
    init_marker_pos: tuple = ([[]], [[]])
    yaw_source: Callable = None
    """Callable returning the (num_envs,) yaw [rad] of the indenter relative to the sensor; replaces the
    FrameTransformer read-out of fots_marker_sim.py:147-159."""
