#!/usr/bin/env python3
"""GPU side of tests/studies/shadow_outliers.py: the HIP path's deformed gel, contact mask and shadow-branch RGB of the 240x320
fixture, written to gpurun_out/shadow_dump.npz for the attribution done in the container (where the reference can be imported)."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim  # noqa: E402

g = dict(np.load(REPO / "tests/golden/taxim_240x320.npz"))
calib = REPO / "tacex_amd/assets/calib/gsmini_640x480"
t = Taxim(calib_folder=calib, backend="hip", device="cuda:0")
hm, indent = torch.from_numpy(g["hm"]).cuda(), torch.from_numpy(g["indent"]).cuda()
rgb = t.render_direct(hm, with_shadow=True, press_depth=indent).movedim(1, 3).cpu().numpy()
Z, M = t.deform(hm, indent)
_, idx = t.shade(Z, return_bins=True)
out = REPO / "gpurun_out"
out.mkdir(exist_ok=True)
np.savez_compressed(out / "shadow_dump.npz", rgb=rgb, Z=Z.cpu().numpy(), M=np.packbits(M.cpu().numpy().astype(bool)), idx=idx.cpu().numpy().astype(np.uint8))
print("written", (out / "shadow_dump.npz").stat().st_size // 1024, "KiB")
