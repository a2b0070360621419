"""Is a sensor of 2n envs equal to two sensors of n envs on the same frames (markers, trajectory state)?  For n = 2 .. 8."""
import sys, torch
sys.path.insert(0, "tests")
from test_sensor_gpu import make_sensor
from tacex_amd.calibration import CALIB_GELSIGHT_MINI as calib
from tacex_amd.utils.synthetic import synthetic_depth_maps
for n in (2, 3, 4, 5, 6, 8):
    hm = [synthetic_depth_maps(n, 240, 320, seed=41 + k, flat_fraction=0.0)[0].cuda() / 1000.0 for k in range(2)]
    small = [make_sensor(calib, n), make_sensor(calib, n)]
    big = make_sensor(calib, 2 * n)
    for s in small + [big]:
        s.initialize()
    for k in range(2):
        small[k].set_camera_depth(hm[k]); small[k].update(dt=0.01, force_recompute=True)
    big.set_camera_depth(torch.cat(hm)); big.update(dt=0.01, force_recompute=True)
    mk = torch.cat([s.data.output["marker_motion"] for s in small]); tr = torch.cat([s.marker_motion_simulator._traj_state for s in small])
    rgb = torch.cat([s.data.output["tactile_rgb"] for s in small])
    o = big.data.output
    d = (mk - o["marker_motion"]).abs().amax(dim=(1, 2, 3)).cpu().tolist()
    print(n, "rgb equal", torch.equal(rgb, o["tactile_rgb"]), "marker diff per frame", [round(v, 3) for v in d],
          "n_contacts small", tr[:, 7].cpu().tolist(), "big", big.marker_motion_simulator._traj_state[:, 7].cpu().tolist())
    pm_s = torch.cat([s.optical_simulator._pix_m for s in small]).sum(1).cpu().tolist(); pm_b = big.optical_simulator._pix_m.sum(1).cpu().tolist()
    print("   pix_m sums small", pm_s, "big", pm_b)
