"""Diagnostic: per-level deformed-gel comparison HIP vs oracle with a truncated pyramid (1..3 levels).  GPU only."""
import json, os, sys, tempfile
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.calibration import CALIB_GELSIGHT_MINI
from tacex_amd.simulation_approaches.gpu_taxim.sim import TaximHip
from tacex_amd.utils.synthetic import synthetic_depth_maps
from oracle.taxim_oracle import TaximOracle

def calib_with(nlev):
    d = Path(tempfile.mkdtemp())
    for f in CALIB_GELSIGHT_MINI.iterdir():
        if f.name != "params.json":
            os.symlink(f, d / f.name)
    p = json.load(open(CALIB_GELSIGHT_MINI / "params.json"))
    s = p["simulator"]
    s["deform_pyramid_sigma_rel"] = [s["deform_pyramid_sigma_rel"][0][:nlev], s["deform_pyramid_sigma_rel"][1][:nlev]]
    s["deform_final_sigma_rel"] = [1e-7, 1e-7]
    json.dump(p, open(d / "params.json", "w"))
    return d

shape = (240, 320)
hm, _ = synthetic_depth_maps(2, *shape, seed=11, flat_fraction=0.0)
for nlev in (1, 2, 3):
    cd = calib_with(nlev)
    o = TaximOracle(cd, shape, "direct")
    ind = o.indentation_depth(hm.numpy())
    Zo, Mo = o.gel_pad_deformation(o.shifted_height_map(hm.numpy(), ind))
    t = TaximHip(cd, device="cuda:0")
    Z, M = t.deform(hm.cuda(), torch.from_numpy(ind).cuda())
    d = np.abs(Z.cpu().numpy() - Zo)
    am = np.unravel_index(d.argmax(), d.shape)
    print("levels", nlev, "maxdiff %.3e" % d.max(), "argmax", am, "got", Z.cpu().numpy()[am], "want", Zo[am], "mask ne", int((M.cpu().numpy().astype(bool) != Mo).sum()))
    bad = d[0] > 1e-5
    ys, xs = np.where(bad)
    if len(ys):
        print("   bad px", bad.sum(), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "row%32 hist", np.bincount(ys % 32, minlength=32).tolist())
