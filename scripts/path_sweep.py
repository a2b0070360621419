"""Consistency sweep (GPU): the default fast path (MFMA levels, fused tail, fused observation, FOTS by-products) against the
unfused per-level path, for several shard sizes and both resolutions.  Prints the differences of RGB, markers, observation."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tacex_amd import GelSightSensor, GelSightSensorCfg
from tacex_amd.calibration import CALIB_GELSIGHT_MINI
from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg
from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg
from tacex_amd.utils.synthetic import synthetic_depth_maps

def make(n, W, H, obs_dtype):
    cfg = GelSightSensorCfg(num_envs=n, data_types=["tactile_rgb", "marker_motion", "height_map"],
        sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(CALIB_GELSIGHT_MINI), gelpad_height=0.0045,
            gelpad_to_camera_min_distance=0.024, tactile_img_res=(W, H), device="cuda:0", policy_obs_res=(32, 32), policy_obs_dtype=obs_dtype),
        marker_motion_sim_cfg=FOTSMarkerSimulatorCfg(tactile_img_res=(W, H), device="cuda:0"), device="cuda:0")
    return GelSightSensor(cfg)

for (W, H), n in (((320, 240), 1), ((320, 240), 7), ((320, 240), 300), ((640, 480), 3), ((640, 480), 70)):
    hm, _ = synthetic_depth_maps(n, H, W, seed=n)
    res = []
    for mode in ("fast", "safe"):
        s = make(n, W, H, "float32")
        s.initialize()
        if mode == "safe":  # separate kernels per level, FOTS from full frames via its own reduction
            s.optical_simulator._taxim.set_fused_tail((H, W), False)
        for k in range(2):
            s.set_camera_depth((hm / 1000.0).cuda() + 1e-5 * k)
            s.update(0.01, force_recompute=True)
        o = s.data.output
        res.append((o["tactile_rgb"].cpu().numpy(), o["marker_motion"].cpu().numpy(), o["tactile_rgb_obs"].cpu().numpy(),
                    s.optical_simulator._fots_compact_version))
    a, b = res
    drgb = np.abs(a[0] - b[0]); dm = np.abs(a[1] - b[1]).max(); dobs = np.abs(a[2] - b[2]).max()
    print(f"{W}x{H} n={n}: compact {a[3] >= 0}/{b[3] >= 0}  rgb mean|d| {drgb.mean():.2e} q99.9 {np.quantile(drgb, 0.999):.2e}  markers max|d| {dm:.2e}  obs max|d| {dobs:.2e}")
