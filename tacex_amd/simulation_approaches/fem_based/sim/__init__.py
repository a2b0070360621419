from .tactile_sensor_uipc import VisionTactileSensorUIPC

__all__ = ["VisionTactileSensorUIPC"]
