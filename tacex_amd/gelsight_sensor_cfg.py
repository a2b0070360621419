from typing import Callable, Literal

from .gelsight_sensor import GelSightSensor
from .sensor_base import SensorBaseCfg
from .simulation_approaches.gelsight_simulator_cfg import GelSightSimulatorCfg
from .utils.configclass import configclass


@configclass
class GelSightSensorCfg(SensorBaseCfg):
    """Same fields as the reference's gelsight_sensor_cfg.py:12-64 (+ the depth source that replaces TiledCamera)."""

    class_type: type = GelSightSensor

    @configclass
    class Dimensions:
        """Dimensions in metres."""

        width: float = 0.0
        length: float = 0.0
        height: float = 0.0

    case_dimensions: Dimensions = Dimensions()
    gelpad_dimensions: Dimensions = Dimensions()

    @configclass
    class SensorCameraCfg:
        """Configs for the camera of the GelSight sensor."""

        prim_path_appendix: str = "/Camera"
        update_period: float = 0
        resolution: tuple = (32, 24)
        data_types: list = ["depth"]
        clipping_range: tuple = (0.0, 1.0)
        depth_source: Callable = None
        """Callable returning the camera depth image in metres, (num_envs, H, W) or (num_envs, H, W, 1), on the
        sensor's device - what `TiledCamera.data.output["depth"]` provides in the reference
        (gelsight_sensor.py:229-263, 581-593).  Alternatively call `sensor.set_camera_depth(...)`."""

    sensor_camera_cfg: SensorCameraCfg = SensorCameraCfg()

    data_types: list = ["tactile_rgb", "marker_motion", "height_map", "camera_depth", "camera_rgb"]
    optical_sim_cfg: GelSightSimulatorCfg = None
    marker_motion_sim_cfg: GelSightSimulatorCfg = None
    compute_indentation_depth_class: Literal["optical_sim", "marker_motion_sim"] = "optical_sim"
    device: str = "cuda"
    reset_gelpad_with_sensor: bool = True
    """Not in the reference cfg: `GelSightSensor.reset(env_ids)` also puts the FEM gelpad (`gelpad_obj`, a tacex_uipc UipcObject) of those
    envs back to rest (`UipcObject.reset`).  False: the caller resets the pad itself."""
