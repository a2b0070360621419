from ...utils.configclass import configclass
from ..gelsight_simulator_cfg import GelSightSimulatorCfg
from .mani_skill_sim import ManiSkillSimulator


@configclass
class ManiSkillSimulatorCfg(GelSightSimulatorCfg):
    """Same fields as the reference's fem_based/mani_skill_sim_cfg.py:9-70 (+ the camera pose IsaacLab's camera supplies)."""

    simulation_approach_class: type = ManiSkillSimulator
    calib_folder_path: str = ""
    device: str = "cuda"
    marker_interval_range: tuple = (2.0625, 2.0625)
    marker_rotation_range: float = 0.0
    marker_translation_range: tuple = (0.0, 0.0)
    marker_pos_shift_range: tuple = (0.0, 0.0)
    marker_random_noise: float = 0.0
    marker_lose_tracking_probability: float = 0.0
    normalize: bool = False
    marker_flow_size: int = 128
    camera_params: tuple = (340, 325, 160, 125, 0.0)
    tactile_img_res: tuple = (320, 240)
    camera_pos_w: tuple = (0.0, 0.0, 0.0)
    """World position of the sensor camera (the reference reads it from the TiledCamera, VT:158-160)."""
    camera_quat_w_ros: tuple = (1.0, 0.0, 0.0, 0.0)
    """World orientation (w,x,y,z) of the camera in the ROS / OpenCV convention."""

    @configclass
    class MarkerParams:
        num_markers: int = 128
        x0: float = 0
        y0: float = 0
        dx: float = 0
        dy: float = 0

    marker_params: MarkerParams = MarkerParams()
    init_marker_pos: tuple = ([[]], [[]])
