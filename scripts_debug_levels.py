import json, os, sys, tempfile, shutil
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, '.')
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

for shape in [(32, 32), (24, 32), (48, 64), (240, 320)]:
    H, W = shape
    hm, _ = synthetic_depth_maps(3, H, W, seed=11, flat_fraction=0.0)
    for nlev in range(1, 7):
        cd = calib_with(nlev)
        o = TaximOracle(cd, shape, "direct")
        ind = o.indentation_depth(hm.numpy())
        Zo, Mo = o.gel_pad_deformation(o.shifted_height_map(hm.numpy(), ind))
        t = TaximHip(cd, device="cuda:0")
        Z, M = t.deform(hm.cuda(), torch.from_numpy(ind).cuda())
        tb = t.context(shape).tables
        d = np.abs(Z.cpu().numpy() - Zo)
        print(shape, "levels", nlev, "k", list(zip(tb.ksize_w, tb.ksize_h)), "maxdiff %.3e" % d.max(), "argmax", np.unravel_index(d.argmax(), d.shape), "mask ne", int((M.cpu().numpy().astype(bool) != Mo).sum()))
