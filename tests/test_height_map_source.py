"""Analytic-indenter height-map source: oracle vs the synthetic generator (CPU), HIP kernel vs oracle and
the sensor path vs the camera-depth path (GPU)."""
import numpy as np
import pytest
import torch


def _desc_like_synthetic(B, H, W, seed, kinds=("sphere", "cylinder", "edge", "two_spheres")):
    """The indenter parameters `synthetic_depth_maps` draws for (B, H, W, seed) with flat_fraction = 0."""
    import math

    g = torch.Generator().manual_seed(seed)
    u = torch.rand((B, 8), generator=g)
    kind_id = torch.randint(0, len(kinds), (B,), generator=g)
    _ = torch.rand((B,), generator=g)
    code = {"sphere": 0.0, "cylinder": 1.0, "edge": 2.0, "two_spheres": 3.0}
    d = torch.zeros((B, 8))
    d[:, 0] = torch.tensor([code[kinds[int(k)]] for k in kind_id])
    r = (0.15 + 0.20 * u[:, 1]) * H
    d[:, 1] = (0.3 + 0.4 * u[:, 2]) * W
    d[:, 2] = (0.3 + 0.4 * u[:, 3]) * H
    d[:, 3] = r
    d[:, 4] = math.pi * u[:, 4]
    d[:, 5] = 0.2 + 1.3 * u[:, 0]
    d[:, 6] = d[:, 1] + (0.5 + u[:, 5]) * r
    d[:, 7] = d[:, 2] + (u[:, 6] - 0.5) * r
    return d


def test_oracle_matches_the_synthetic_generator():
    """Pins the oracle's scene model to the generator the golden vectors were made with."""
    from oracle.indenter_oracle import indenter_height_map, indentation_depth
    from tacex_amd.utils.synthetic import PIXMM, synthetic_depth_maps

    B, H, W = 12, 240, 320
    ref, ind = synthetic_depth_maps(B, H, W, seed=5, flat_fraction=0.0)
    got = indenter_height_map(_desc_like_synthetic(B, H, W, 5).numpy(), H, W, PIXMM)
    assert np.abs(got - ref.numpy()).max() <= 2e-5
    m, dep = indentation_depth(got)
    assert np.abs(m - got.reshape(B, -1).min(1)).max() == 0 and (dep > 0).all()
    flat = indenter_height_map(np.array([[-1, 0, 0, 0, 0, 0, 0, 0]], np.float32), 24, 32, PIXMM)
    assert (flat == np.float32(29.0)).all()


@pytest.mark.gpu
def test_hip_source_vs_oracle_and_sensor_path(calib_dir):
    from oracle.indenter_oracle import indenter_height_map, indentation_depth
    from tacex_amd import GelSightSensor, GelSightSensorCfg, IndenterHeightMapSource
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg
    from tacex_amd.utils.synthetic import PIXMM

    B, H, W = 9, 240, 320
    desc = _desc_like_synthetic(B, H, W, 21)
    desc[3, 0] = -1.0  # one env without contact
    want = indenter_height_map(desc.numpy(), H, W, PIXMM)
    fmin_o, ind_o = indentation_depth(want)

    def make():
        cfg = GelSightSensorCfg(
            num_envs=B, data_types=["tactile_rgb", "height_map"],
            sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
            optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(calib_dir), gelpad_height=0.0045,
                                              gelpad_to_camera_min_distance=0.024, tactile_img_res=(W, H), device="cuda"),
            device="cuda")
        return GelSightSensor(cfg)

    s = make()
    src = IndenterHeightMapSource(B, "cuda", pixmm=PIXMM)
    src.params.copy_(desc)
    s.set_height_map_source(src)
    s.update(0.01, force_recompute=True)
    hm = s.data.output["height_map"].cpu().numpy()
    assert np.abs(hm - want).max() <= 2e-5  # float32 sqrt / sincos round-off
    np.testing.assert_allclose(s.indentation_depth.cpu().numpy(), ind_o, atol=3e-5)
    assert float(s.indentation_depth[3]) == 0.0
    # the same scene through the camera-depth path renders the same tactile frame
    s2 = make()
    s2.set_camera_depth(torch.from_numpy(want / 1000.0).cuda())
    s2.update(0.01, force_recompute=True)
    a, b = s.data.output["tactile_rgb"], s2.data.output["tactile_rgb"]
    assert (a - b).abs().mean().item() < 1e-4  # m -> mm round trip of the depth path moves a few flat-gel bins
