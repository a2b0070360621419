import json, os, sys, tempfile
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.calibration import CALIB_GELSIGHT_MINI
from tacex_amd.simulation_approaches.gpu_taxim.sim import TaximHip
from tacex_amd.utils.synthetic import synthetic_depth_maps
d = Path(tempfile.mkdtemp())
for f in CALIB_GELSIGHT_MINI.iterdir():
    if f.name != "params.json":
        os.symlink(f, d / f.name)
p = json.load(open(CALIB_GELSIGHT_MINI / "params.json")); s = p["simulator"]
s["deform_pyramid_sigma_rel"] = [s["deform_pyramid_sigma_rel"][0][:1], s["deform_pyramid_sigma_rel"][1][:1]]
s["deform_final_sigma_rel"] = [1e-7, 1e-7]
json.dump(p, open(d / "params.json", "w"))
hm, ind = synthetic_depth_maps(1, 240, 320, seed=11, flat_fraction=0.0)
t = TaximHip(d, device="cuda:0")
Z, M = t.deform(hm.cuda(), ind.cuda())
np.save(sys.argv[1], Z.cpu().numpy()); np.save(sys.argv[1] + ".m.npy", M.cpu().numpy())
