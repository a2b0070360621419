from ...utils.configclass import MISSING, configclass
from ..gelsight_simulator_cfg import GelSightSimulatorCfg
from .taxim_sim import TaximSimulator


@configclass
class TaximSimulatorCfg(GelSightSimulatorCfg):
    """Same fields as the reference's gpu_taxim/taxim_sim_cfg.py:11-36."""

    simulation_approach_class: type = TaximSimulator
    calib_folder_path: str = ""
    device: str = "cuda"
    with_shadow: bool = False
    tactile_img_res: tuple = (320, 240)
    """(width, height) of the tactile image; the camera height map is resampled if it differs."""
    gelpad_height: float = MISSING
    """Used for computing the indentation depth from the height map [m]."""
    gelpad_to_camera_min_distance: float = MISSING
    """Min distance of the camera to the gelpad [m]."""
    policy_obs_res: tuple = None
    """Extension (not in the reference): (width, height) of an antialiased low-resolution copy of the tactile frame
    produced in the same pass (e.g. (32, 32), what the TacEx tasks feed to the policy); exposed as
    `sensor.data.output["tactile_rgb_obs"]` with shape (num_envs, height, width, 3)."""
    policy_obs_dtype: str = "float32"
    """"float32" (values in [0,1]) or "uint8" (floor(255 x + 0.5): the image a CNN policy consumes, a quarter of the bytes
    in the per-step observation gather)."""
