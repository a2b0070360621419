from dataclasses import dataclass
from typing import Any

import numpy as np


@dataclass
class GelSightSensorData:
    """Data container for a GelSight sensor (reference: gelsight_sensor_data.py:6-23)."""

    position: np.ndarray = None
    orientation: np.ndarray = None
    intrinsic_matrix: np.ndarray = None
    image_resolution: tuple = None
    output: dict[str, Any] = None
    """Sensor outputs keyed by data type: "camera_depth", "height_map", "tactile_rgb", "marker_motion"."""
