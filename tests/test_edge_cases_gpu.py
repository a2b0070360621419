"""GPU edge cases of the Taxim / sensor boundary: ragged and degenerate inputs the reference accepts."""
import numpy as np
import pytest
import torch

from parity import rgb_rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def taxim(calib_dir):
    from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim

    return Taxim(calib_folder=calib_dir, backend="hip", device="cuda:0")


@pytest.fixture(scope="module")
def oracle(calib_dir):
    from oracle.taxim_oracle import TaximOracle

    return TaximOracle(calib_dir, (240, 320), "direct")


def assert_parity(taxim, oracle, hm, ind, out_nhwc):
    """Same-bin protocol: bins equal on >= 99 % of strong-gradient pixels, RGB <= 1e-4 rel on same-bin pixels."""
    S = oracle.shifted_height_map(hm, ind)
    Zo, _ = oracle.gel_pad_deformation(S)
    ref, mag, _, im, idd = oracle.shade(Zo, True)
    Z, _ = taxim.deform(torch.from_numpy(hm).cuda(), torch.from_numpy(np.asarray(ind, np.float32)).cuda())
    assert np.abs(Z.cpu().numpy() - Zo).max() <= 1e-5
    _, idx = taxim.shade(Z, return_bins=True)
    idx = idx.cpu().numpy().astype(np.int64)
    same = (idx[..., 0] == im) & (idx[..., 1] == idd)
    strong = mag > 1e-3
    if strong.any():
        assert same[strong].mean() >= 0.99
    assert rgb_rel_err(out_nhwc, ref)[same].max() <= 1e-4


def _inputs(n, seed, **kw):
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    return synthetic_depth_maps(n, 240, 320, seed=seed, **kw)


def test_single_frame_and_odd_batches(taxim, oracle):
    for n in (1, 3, 7):
        hm, _ = _inputs(n, 200 + n, flat_fraction=0.0)
        ind = oracle.indentation_depth(hm.numpy())
        out = taxim.render_direct(hm.cuda(), False, torch.from_numpy(ind).cuda()).movedim(1, 3).cpu().numpy()
        assert_parity(taxim, oracle, hm.numpy(), ind, out)


def test_batch_dims_and_numpy_entry(taxim, oracle):
    """(E, S, H, W) batch dims are flattened and restored (TT:182-183,194); render() accepts NumPy (TT:166-171)."""
    hm, _ = _inputs(6, 31, flat_fraction=0.0)
    ind = oracle.indentation_depth(hm.numpy())
    out = taxim.render_direct(hm.reshape(3, 2, 240, 320).cuda(), False, torch.from_numpy(ind).cuda())
    assert out.shape == (3, 2, 3, 240, 320)
    flat = taxim.render_direct(hm.cuda(), False, torch.from_numpy(ind).cuda())
    assert torch.equal(out.reshape(6, 3, 240, 320), flat)
    arr = taxim.render(hm[:2].numpy(), with_shadow=False, press_depth=torch.from_numpy(ind[:2]).cuda())
    assert isinstance(arr, np.ndarray) and arr.shape == (2, 240, 320, 3)
    np.testing.assert_allclose(arr, flat[:2].movedim(1, 3).cpu().numpy(), atol=0)


def test_scalar_press_depth_and_non_contiguous_input(taxim, oracle):
    hm, _ = _inputs(4, 77, flat_fraction=0.0)
    big = torch.zeros((4, 240, 640))
    big[:, :, ::2] = hm
    view = big.cuda()[:, :, ::2]  # non-contiguous view of the same data
    assert not view.is_contiguous()
    a = taxim.render_direct(view, False, 0.7)          # python float press depth (TI:117-151 allows float)
    b = taxim.render_direct(hm.cuda(), False, torch.full((4,), 0.7, device="cuda"))
    assert torch.equal(a, b)
    assert_parity(taxim, oracle, hm.numpy(), np.full(4, 0.7, np.float32), a.movedim(1, 3).cpu().numpy())
    with pytest.raises(ValueError):
        taxim.render_direct(hm.cuda(), False, torch.zeros(3, device="cuda"))  # wrong number of press depths


def test_orig_hm_fmt(taxim, oracle):
    """orig_hm_fmt=True: height_map := gel_map_shift - height_map (TT:185-186)."""
    hm, _ = _inputs(2, 5, flat_fraction=0.0)
    ind = oracle.indentation_depth(hm.numpy())
    shift = taxim.context((240, 320)).tables.gel_map_shift
    assert abs(shift - oracle.gel_map_shift) < 1e-6
    a = taxim.render_direct((shift - hm).cuda(), False, torch.from_numpy(ind).cuda(), orig_hm_fmt=True)
    b = taxim.render_direct(hm.cuda(), False, torch.from_numpy(ind).cuda())
    assert (a - b).abs().max().item() <= 2e-6


def test_empty_batch_and_flat_frames(taxim, oracle):
    out = taxim.render_direct(torch.zeros((0, 240, 320), device="cuda"), False, torch.zeros(0, device="cuda"))
    assert out.shape == (0, 3, 240, 320)
    # all-background frames: exactly background + poly(bin 0, 62), identical for every frame
    hm = torch.full((3, 240, 320), 29.0)
    out = taxim.render_direct(hm.cuda(), False, torch.zeros(3, device="cuda")).movedim(1, 3).cpu().numpy()
    ref = oracle.render_direct(hm.numpy(), np.zeros(3, np.float32))
    assert np.abs(out - ref).max() <= 2e-6
    assert np.array_equal(out[0], out[2])


def test_object_inside_sensor_case_and_extreme_press(taxim, oracle):
    """Depth closer than the sensor case clamps the distance to 0 -> indentation = gelpad height (TS:124-129);
    very deep presses stay finite and in [0, 1]."""
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    hm, _ = synthetic_depth_maps(2, 240, 320, seed=9, flat_fraction=0.0)
    hm[0] -= 5.0  # min ~ 22.5 mm < 24 mm
    ind = oracle.indentation_depth(hm.numpy())
    assert ind[0] == np.float32(4.5)
    out = taxim.render_direct(hm.cuda(), False, torch.from_numpy(ind).cuda())
    assert torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    assert_parity(taxim, oracle, hm.numpy(), ind, out.movedim(1, 3).cpu().numpy())


@pytest.mark.parametrize("shape", [(30, 40), (50, 70), (96, 128)])
def test_unusual_resolutions_generic_path(taxim, calib_dir, shape):
    """Resolutions without tuned kernels (odd sizes, W % 16 != 0) run on the generic kernels and still match."""
    from oracle.taxim_oracle import TaximOracle
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    H, W = shape
    hm, _ = synthetic_depth_maps(3, H, W, seed=H, flat_fraction=0.0)
    o = TaximOracle(calib_dir, shape, "direct")
    ind = o.indentation_depth(hm.numpy())
    Zo, Mo = o.gel_pad_deformation(o.shifted_height_map(hm.numpy(), ind))
    Z, M = taxim.deform(hm.cuda(), torch.from_numpy(ind).cuda())
    assert np.abs(Z.cpu().numpy() - Zo).max() <= 1e-5
    np.testing.assert_array_equal(M.cpu().numpy().astype(bool), Mo)
    out = taxim.render_direct(hm.cuda(), False, torch.from_numpy(ind).cuda()).movedim(1, 3).cpu().numpy()
    assert_parity(taxim, o, hm.numpy(), ind, out)  # same-bin protocol, maximum


def test_params_override_and_unknown_key(calib_dir):
    from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim

    t = Taxim(calib_folder=calib_dir, params={"simulator": {"contact_scale": 0.5}}, backend="hip", device="cuda:0")
    assert t.sim_params.contact_scale == 0.5 and t.backend_name == "hip" and (t.width, t.height) == (640, 480)
    with pytest.raises(ValueError, match="Unknown key"):
        Taxim(calib_folder=calib_dir, params={"simulator": {"bogus": 1}}, backend="hip", device="cuda:0")
    assert t.background_img.shape == (3, 480, 640)


def test_chunked_pipeline_is_bit_identical(calib_dir, tmp_path):
    """Large shards are walked in chunks that re-use the same scratch images (Infinity-Cache residency).  Chunking must
    not change a single bit of RGB, deformed gel, mask or observation (TACEX_CHUNK_FRAMES is read once per process)."""
    import os
    import subprocess
    import sys

    from conftest import REPO

    script = tmp_path / "chunk_run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim\n"
        "from tacex_amd.utils.synthetic import synthetic_depth_maps\n"
        f"t = Taxim(calib_folder={str(calib_dir)!r}, backend='hip', device='cuda:0')\n"
        "hm, ind = synthetic_depth_maps(7, 240, 320, seed=11, device='cuda')\n"
        "z = torch.empty((7, 240, 320), device='cuda'); m = torch.empty((7, 240, 320), dtype=torch.uint8, device='cuda')\n"
        "obs = torch.empty((7, 32, 32, 3), device='cuda')\n"
        "rgb = t.render_direct(hm, False, ind, z_out=z, mask_out=m, obs_out=obs)\n"
        "np.savez(sys.argv[1], rgb=rgb.cpu().numpy(), z=z.cpu().numpy(), m=m.cpu().numpy(), obs=obs.cpu().numpy())\n")
    outs = {}
    for chunk in ("0", "3"):
        out = tmp_path / f"c{chunk}.npz"
        env = dict(os.environ, TACEX_CHUNK_FRAMES=chunk)
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[chunk] = np.load(out)
    for k in ("rgb", "z", "m", "obs"):
        np.testing.assert_array_equal(outs["0"][k], outs["3"][k])


def test_curved_gel_map_general_path(calib_dir, tmp_path):
    """The shipped GelSight Mini gel map is identically zero, which the kernels exploit (no gel loads).  A curved gel pad
    (non-zero map) must take the general path and still match the oracle built from the same calibration folder."""
    import shutil

    from oracle.taxim_oracle import TaximOracle
    from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim

    folder = tmp_path / "calib_curved"
    shutil.copytree(calib_dir, folder)
    gm = np.load(folder / "gelmap.npy").astype(np.float32)
    yy, xx = np.mgrid[0:gm.shape[0], 0:gm.shape[1]].astype(np.float32)
    cy, cx = (gm.shape[0] - 1) / 2, (gm.shape[1] - 1) / 2
    np.save(folder / "gelmap.npy", (-((yy - cy) ** 2 + (xx - cx) ** 2) / 4000.0).astype(np.float32))  # dome, up to ~ -40 px
    tx = Taxim(calib_folder=folder, backend="hip", device="cuda:0")
    orc = TaximOracle(folder, (240, 320), "direct")
    assert np.abs(orc.gel).max() > 0.1 if hasattr(orc, "gel") else True
    hm, _ = _inputs(3, 77, flat_fraction=0.0)
    ind = orc.indentation_depth(hm.numpy())
    out = tx.render_direct(hm.cuda(), False, torch.from_numpy(ind).cuda()).movedim(1, 3).cpu().numpy()
    assert_parity(tx, orc, hm.numpy(), ind, out)


def test_full_baseline_batch_2048_frames_auto_chunked(calib_dir):
    """BASELINE config C3 size: 2048 frames through the sensor boundary take the automatic chunk policy (Infinity-Cache-sized
    passes over shared scratch images).  A 64-env sensor fed the same depth maps renders in ONE pass: RGB, height map, uint8
    policy observation and indentation depth must be bit-equal on that subset; markers follow to float32 round-off (FOTS
    subtracts the BATCH maximum of the deformed gel, FS:130, which differs between the two batches by design)."""
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    def sensor(n):
        cfg = GelSightSensorCfg(
            num_envs=n, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(320, 240), clipping_range=(0.024, 0.029)),
            data_types=["tactile_rgb", "height_map", "marker_motion"],
            optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(calib_dir), gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,
                                              tactile_img_res=(320, 240), device="cuda:0", policy_obs_res=(32, 32), policy_obs_dtype="uint8"),
            marker_motion_sim_cfg=FOTSMarkerSimulatorCfg(tactile_img_res=(320, 240), device="cuda:0"), device="cuda:0")
        s = GelSightSensor(cfg)
        s.initialize()
        return s

    N, lo, n = 3200, 1611, 64  # (one streaming pass holds up to 2048 frames of 320x240; 3200 are walked as two passes of 1600)
    hm, _ = synthetic_depth_maps(N, 240, 320, seed=2048, device="cuda:0")
    depth = (hm / 1000.0).contiguous()
    big = sensor(N)
    assert big.optical_simulator._taxim.chunk_frames((240, 320), N) < N  # the chunk policy really engages at this size
    big.set_camera_depth(depth)
    small = sensor(n)
    assert small.optical_simulator._taxim.chunk_frames((240, 320), n) == n
    small.set_camera_depth(depth[lo:lo + n].clone())
    for _ in range(2):  # second step: FOTS trajectories (shear / twist) are live
        big.update(0.01, force_recompute=True)
        small.update(0.01, force_recompute=True)
    ob, os_ = big.data.output, small.data.output
    assert torch.equal(ob["height_map"][lo:lo + n], os_["height_map"])
    assert torch.equal(big.indentation_depth[lo:lo + n], small.indentation_depth)
    assert torch.equal(ob["tactile_rgb"][lo:lo + n], os_["tactile_rgb"])
    assert torch.equal(ob["tactile_rgb_obs"][lo:lo + n], os_["tactile_rgb_obs"])
    assert float(ob["tactile_rgb_obs"].float().std()) > 1.0
    mb, ms = ob["marker_motion"][lo:lo + n], os_["marker_motion"]
    assert torch.equal(mb[:, 0], ms[:, 0])
    assert float((mb[:, 1] - ms[:, 1]).abs().max()) <= 1e-4
    assert float((mb[:, 1] - mb[:, 0]).abs().max()) > 0.1


def test_policy_observation_with_shadow(taxim):
    """`obs_out` together with `with_shadow=True` (it used to stay all-zero silently): the observation is the antialiased
    down-sample of the shadowed frame, float32 and uint8."""
    hm, ind = _inputs(3, 91, flat_fraction=0.0)
    hm, ind = hm.cuda(), ind.cuda()
    for dt in (torch.float32, torch.uint8):
        obs = torch.zeros((3, 32, 32, 3), dtype=dt, device="cuda")
        rgb = taxim.render_direct(hm, True, ind, obs_out=obs)  # (3, 3, H, W) view
        ref = torch.nn.functional.interpolate(rgb, size=[32, 32], mode="bilinear", antialias=True).movedim(1, 3)
        if dt == torch.uint8:
            q = torch.floor(255.0 * ref + 0.5)
            assert float((obs.float() - q).abs().max()) <= 1.0 and float((obs.float() == q).float().mean()) > 0.99
        else:
            assert float((obs - ref).abs().max()) < 1e-5
        assert float(obs.float().std()) > 0
