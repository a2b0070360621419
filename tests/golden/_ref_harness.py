"""Container-only harness that imports the *reference* Taxim / FOTS kernels from /root/reference.

Used ONLY by the golden-vector generators (make_golden.py & co., run in the build container where /root/reference exists);
the tests themselves (tests/test_oracle_golden.py, tests/test_*_gpu.py) read the committed .npz fixtures, never this module.
Nothing here is product code; no reference source is copied - modules are loaded from where they lie.

Recipe follows SURVEY.md section 8(c):
  * torchvision / torch_scatter / cv2 are absent from the image -> tiny shims in sys.modules
  * gpu_taxim/sim is loaded as a standalone package (its parents import omni / isaaclab)
  * dataPack.npz (background frame f0) is missing from the mount -> synthesized, seeded
"""
from __future__ import annotations

import importlib.util
import shutil
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REF_ROOT = Path("/root/reference")
REF_SIM = REF_ROOT / "source/tacex/tacex/simulation_approaches/gpu_taxim/sim"
REF_FOTS = REF_ROOT / "source/tacex/tacex/simulation_approaches/fots/sim/marker_motion.py"
REF_CALIB = REF_ROOT / "source/tacex_assets/tacex_assets/data/Sensors/GelSight_Mini/calibs/640x480"


def reference_available() -> bool:
    return REF_SIM.is_dir() and REF_CALIB.is_dir()


def _install_shims() -> None:
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvf = types.ModuleType("torchvision.transforms.functional")

        class InterpolationMode:  # only BILINEAR is used by the reference
            BILINEAR = "bilinear"

        def resize(img, size, interpolation="bilinear", antialias=True):
            # torchvision>=0.17 semantics: bilinear + antialias on (C,H,W) / (B,C,H,W) tensors
            x = img
            squeeze = False
            if x.dim() == 3:
                x = x[None]
                squeeze = True
            y = torch.nn.functional.interpolate(
                x, size=list(size), mode="bilinear", align_corners=False, antialias=antialias
            )
            return y[0] if squeeze else y

        tvt.InterpolationMode = InterpolationMode
        tvf.resize = resize
        tvt.functional = tvf
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
        sys.modules["torchvision.transforms.functional"] = tvf
    if "torch_scatter" not in sys.modules:
        ts = types.ModuleType("torch_scatter")

        def scatter_min(src, index, dim_size=None, out=None):
            out.scatter_reduce_(-1, index.expand_as(src), src, "amin", include_self=True)
            return out, None

        ts.scatter_min = scatter_min
        sys.modules["torch_scatter"] = ts
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")


def synth_f0(seed: int = 7) -> np.ndarray:
    """Seeded smooth background frame, 480x640x3, BGR order, 0..255 scale (float64 like np.load gives)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, 480), np.linspace(0, 1, 640), indexing="ij")
    img = np.empty((480, 640, 3), dtype=np.float64)
    for c in range(3):
        a = rng.uniform(90, 150)
        img[..., c] = (
            a
            + 25 * np.sin(2 * np.pi * (rng.uniform(0.3, 1.2) * xx + rng.uniform(0, 1)))
            + 18 * np.cos(2 * np.pi * (rng.uniform(0.3, 1.2) * yy + rng.uniform(0, 1)))
            + 12 * xx * yy
        )
    img += rng.normal(0, 1.5, img.shape)
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


def make_calib_dir(dst: Path, seed: int = 7) -> Path:
    """Build a calibration folder: the reference's real tables + the synthesized dataPack.npz."""
    dst.mkdir(parents=True, exist_ok=True)
    for name in ("params.json", "polycalib.npz", "gelmap.npy", "shadowTable.npz"):
        shutil.copyfile(REF_CALIB / name, dst / name)
    np.savez_compressed(dst / "dataPack.npz", f0=synth_f0(seed))
    return dst


_loaded = {}


def load_reference_taxim(calib_dir: Path | None = None):
    """Returns (TaximTorch instance on cpu, module). Private stages are reachable name-mangled."""
    _install_shims()
    if "pkg" not in _loaded:
        name = "_ref_gpu_taxim_sim"
        spec = importlib.util.spec_from_file_location(
            name, REF_SIM / "__init__.py", submodule_search_locations=[str(REF_SIM)]
        )
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        _loaded["pkg"] = mod
    mod = _loaded["pkg"]
    if calib_dir is None:
        calib_dir = make_calib_dir(Path(tempfile.mkdtemp(prefix="tacex_calib_")))
    t = mod.TaximTorch(calib_folder=Path(calib_dir), device="cpu")
    return t, mod


def load_reference_marker_motion():
    _install_shims()
    if "mm" not in _loaded:
        spec = importlib.util.spec_from_file_location("_ref_marker_motion", REF_FOTS)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _loaded["mm"] = mod
    return _loaded["mm"].MarkerMotion
