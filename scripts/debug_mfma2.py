import json, os, sys, tempfile
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.calibration import CALIB_GELSIGHT_MINI, gaussian_taps
from tacex_amd.simulation_approaches.gpu_taxim.sim import TaximHip
d = Path(tempfile.mkdtemp())
for f in CALIB_GELSIGHT_MINI.iterdir():
    if f.name != "params.json":
        os.symlink(f, d / f.name)
p = json.load(open(CALIB_GELSIGHT_MINI / "params.json")); s = p["simulator"]
s["deform_pyramid_sigma_rel"] = [s["deform_pyramid_sigma_rel"][0][:1], s["deform_pyramid_sigma_rel"][1][:1]]
s["deform_final_sigma_rel"] = [1e-7, 1e-7]
json.dump(p, open(d / "params.json", "w"))
hm = torch.zeros((1, 240, 320)); hm[0, 100, 150] = -1.0
t = TaximHip(d, device="cuda:0")
Z, M = t.deform(hm.cuda(), None)   # no shift: S = hm
Z = Z.cpu().numpy()[0]
w = gaussian_taps(15.25, 61).astype(np.float64)
exp = -np.outer(w, w)
got = Z[70:131, 120:181]
np.set_printoptions(precision=5, linewidth=220, suppress=True)
print("center got", Z[100, 150], "sum got", Z.sum() - Z[100, 150], "sum exp", exp.sum() - exp[30, 30])
r = got / exp
print("ratio along row 100 (every 4th):", r[30, ::4])
print("ratio along col 150 (every 4th):", r[::4, 30])
print("nonzero extent rows", np.where(np.abs(Z).max(1) > 1e-9)[0][[0, -1]], "cols", np.where(np.abs(Z).max(0) > 1e-9)[0][[0, -1]])
