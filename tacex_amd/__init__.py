"""tacex_amd - MI355X-native GelSight tactile-image engine behind TacEx's sensor / simulator plugin API."""
from .gelsight_sensor import GelSightSensor
from .gelsight_sensor_cfg import GelSightSensorCfg
from .gelsight_sensor_data import GelSightSensorData
from .gelsight_sensor_group import GelSightSensorGroup
from .height_map_source import IndenterHeightMapSource, MeshDepthSource

__all__ = ["GelSightSensor", "GelSightSensorCfg", "GelSightSensorData", "GelSightSensorGroup", "IndenterHeightMapSource", "MeshDepthSource"]
