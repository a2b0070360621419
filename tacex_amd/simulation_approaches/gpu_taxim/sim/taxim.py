"""`Taxim(...)` factory - same call signature as the reference's (gpu_taxim/sim/taxim.py:76-107).

Backends: "hip" (this package), "auto" (= hip).  The reference's "torch"/"jax" backends are not part of
this build; asking for them raises ImportError like the reference does for a missing backend
(taxim.py:33-34), any other name raises ValueError (taxim.py:107).
"""
from __future__ import annotations

from pathlib import Path
from typing import Any

from ....calibration import CALIB_GELSIGHT
from .taxim_hip import TaximHip


def Taxim(
    calib_folder: Path = CALIB_GELSIGHT,
    params: dict[str, dict[str, Any]] | None = None,
    backend: str = "auto",
    device: str | None = None,
) -> TaximHip:
    if backend in ("auto", "hip"):
        return TaximHip(calib_folder=Path(calib_folder), params=params, device=device or "cuda")
    if backend in ("torch", "jax"):
        raise ImportError(f"The '{backend}' backend of the reference is not shipped with tacex_amd; use backend='hip'.")
    raise ValueError(f"Unknown backend {backend}")
