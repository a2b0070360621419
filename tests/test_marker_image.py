"""FOTS marker image + RGB x marker overlay (fots_marker_sim.py:346-384, 265-272): integer / byte work, bit-exact.
Golden images come from the reference's own `draw_markers` (tests/golden/make_marker_image_golden.py) run on a seeded synthetic
patch table (the table is an INPUT of the stamping; the reference draws its own with cv2)."""
import numpy as np
import pytest
import torch

from oracle.fots_oracle import draw_markers, marker_overlay, synthetic_patch_table


@pytest.fixture(scope="module")
def golden(golden_dir):
    return dict(np.load(golden_dir / "fots_marker_image.npz"))


@pytest.mark.parametrize("shape", [(240, 320), (480, 640)])
def test_oracle_draw_markers_vs_reference(golden, shape):
    H, W = shape
    d = synthetic_patch_table(int(golden["table_seed"]))
    uv = golden[f"uv_{H}x{W}"]
    for k in range(len(uv)):
        np.testing.assert_array_equal(draw_markers(uv[k], d, 3, W, H), golden[f"img_{H}x{W}"][k])
        np.testing.assert_array_equal(draw_markers(uv[k], d, 4.2, W, H), golden[f"img_size42_{H}x{W}"][k])
    assert (golden[f"img_{H}x{W}"] != 255).mean() > 0.01  # markers really drawn


def test_patch_array_stand_in_and_io(tmp_path):
    """The NumPy stand-in for the cv2-drawn table: right shape, dark dot on white ground, dots grow with the size slot, the
    sub-pixel phase moves the dot; .npz round trip is exact."""
    from tacex_amd.simulation_approaches.fots.marker_patches import generate_patch_array, load_patch_array, save_patch_array

    d = generate_patch_array(10, _phases=[(0, 0), (9, 0)])  # two of the 100 phases keep the test quick
    pa = d["patch_array"]
    assert pa.shape == (10, 10, 50, 12, 12) and pa.dtype == np.uint8
    assert pa[0, 0, 15, 6, 6] < 60 and pa[0, 0, 15, 0, 0] > 200
    dark = (255 - pa[0, 0].astype(int)).sum((1, 2))
    assert (np.diff(dark) >= 0).all() and dark[30] > 2 * dark[0]
    cx = lambda p: ((255 - p.astype(float)) * np.arange(12)[None, :]).sum() / (255 - p.astype(float)).sum()
    assert cx(pa[9, 0, 15]) > cx(pa[0, 0, 15]) + 0.5  # 9/10 of a pixel to the right
    save_patch_array(tmp_path / "p.npz", d)
    d2 = load_patch_array(tmp_path / "p.npz")
    np.testing.assert_array_equal(d2["patch_array"], pa)
    assert d2["super_resolution_ratio"] == 10 and d2["base_circle_radius"] == 1.5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(240, 320), (480, 640)])
def test_marker_image_kernel_vs_reference(golden, calib_dir, shape):
    """All marker sets of the fixture as ONE batch through tacex_fots_marker_image: image bit-equal to the reference's, overlay
    bit-equal to the NumPy statement of FS:268-272."""
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg

    H, W = shape
    uv = golden[f"uv_{H}x{W}"]
    n = len(uv)
    cfg = GelSightSensorCfg(
        num_envs=n, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
        data_types=["tactile_rgb", "height_map", "marker_motion"],
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(calib_dir), gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,
                                          tactile_img_res=(W, H), device="cuda:0"),
        marker_motion_sim_cfg=FOTSMarkerSimulatorCfg(tactile_img_res=(W, H), device="cuda:0",
            marker_params=FOTSMarkerSimulatorCfg.MarkerParams(num_markers_col=11, num_markers_row=9, x0=15 * W // 320, y0=26 * H // 240)),
        device="cuda:0")
    s = GelSightSensor(cfg)
    s.initialize()
    fots = s.marker_motion_simulator
    d = synthetic_patch_table(int(golden["table_seed"]))
    fots.set_patch_array(d)
    md = torch.zeros((n, 2, uv.shape[1], 2), device="cuda:0")
    md[:, 1] = torch.from_numpy(uv).cuda()
    rgb = torch.rand((n, H, W, 3), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
    for size, key in ((3, "img"), (4.2, "img_size42")):
        img, ov = fots.marker_images(md, marker_size=size, overlay_rgb=rgb)
        np.testing.assert_array_equal(img.cpu().numpy(), golden[f"{key}_{H}x{W}"])
        ref_ov = np.stack([marker_overlay(rgb[k].cpu().numpy(), golden[f"{key}_{H}x{W}"][k]) for k in range(n)])
        np.testing.assert_array_equal(ov.cpu().numpy(), ref_ov)
    # the reference's single-sensor signature
    np.testing.assert_array_equal(fots.draw_markers(uv[1], 3, W, H), golden[f"img_{H}x{W}"][1])
    # through the sensor: the markers of a real update, overlaid on the rendered frame
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    hm, _ = synthetic_depth_maps(n, H, W, seed=8, flat_fraction=0.0)
    s.set_camera_depth((hm / 1000.0).cuda())
    s.update(0.01, force_recompute=True)
    out = s.data.output
    img, ov = fots.marker_images(marker_size=3, overlay_rgb=out["tactile_rgb"])
    mm = out["marker_motion"].cpu().numpy()
    for k in range(n):
        ref = draw_markers(mm[k, 1], d, 3, W, H)
        np.testing.assert_array_equal(img[k].cpu().numpy(), ref)
        np.testing.assert_array_equal(ov[k].cpu().numpy(), marker_overlay(out["tactile_rgb"][k].cpu().numpy(), ref))


@pytest.mark.gpu
def test_marker_image_argument_errors():
    from tacex_amd import _lib

    lib = _lib.load_library()
    t = torch.zeros(16, device="cuda:0")
    assert lib.tacex_fots_marker_image(t.data_ptr(), t.data_ptr(), 10, 50, 50, 0, t.data_ptr(), 0, 1, 1, 240, 320, 0) == 2  # slot out of range
    assert lib.tacex_fots_marker_image(t.data_ptr(), t.data_ptr(), 10, 50, 15, 0, 0, t.data_ptr(), 1, 1, 240, 320, 0) == 2  # overlay without rgb
