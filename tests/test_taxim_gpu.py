"""GPU parity tests of the Taxim optical path: HIP kernels (through the C ABI) vs the golden vectors produced by
the reference, and vs the deterministic CPU oracle on fresh seeded inputs."""
import numpy as np
import pytest
import torch

from parity import assert_same_bin_vs_oracle, check_against_reference, rgb_rel_err, unpack_mask

pytestmark = pytest.mark.gpu

SHAPES = [(32, 32), (24, 32), (48, 64), (240, 320), (480, 640)]


@pytest.fixture(scope="module")
def taxim(calib_dir):
    from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim

    return Taxim(calib_folder=calib_dir, backend="hip", device="cuda:0")


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("shape", SHAPES)
def test_hip_vs_reference_golden(taxim, golden_dir, shape):
    H, W = shape
    g = dict(np.load(golden_dir / f"taxim_{H}x{W}.npz"))
    hm = torch.from_numpy(g["hm"]).cuda()
    indent = torch.from_numpy(g["indent"]).cuda()
    Z, M = taxim.deform(hm, indent)
    np.testing.assert_array_equal(_np(M).astype(bool), unpack_mask(g["M"], tuple(M.shape)))
    rgb, idx = taxim.shade(Z, return_bins=True)
    idx = _np(idx).astype(np.int64)
    stats = check_against_reference(_np(Z), idx[..., 0], idx[..., 1], _np(rgb), g)
    print(shape, stats)
    # the one-call render gives exactly the staged result
    out = taxim.render_direct(hm, with_shadow=False, press_depth=indent)
    assert out.shape == (hm.shape[0], 3, H, W)
    # fused-tail shading vs the stand-alone shade kernel: same arithmetic, FMA contraction may differ per kernel
    torch.testing.assert_close(out.movedim(1, 3), rgb, rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape", [(240, 320), (48, 64), (480, 640)])
def test_hip_vs_oracle_all_pixels(taxim, calib_dir, shape):
    """Against the deterministic oracle every pixel is compared, flat regions included."""
    from oracle.taxim_oracle import TaximOracle
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    H, W = shape
    n = 6 if H <= 240 else 2
    hm, _ = synthetic_depth_maps(n, H, W, seed=100 + H, flat_fraction=0.2)
    o = TaximOracle(calib_dir, shape, "direct")
    indent = o.indentation_depth(hm.numpy())
    S = o.shifted_height_map(hm.numpy(), indent)
    Zo, Mo = o.gel_pad_deformation(S)
    rgbo, mago, diro, imo, ido = o.shade(Zo, True)
    Z, M = taxim.deform(hm.cuda(), torch.from_numpy(indent).cuda())
    assert np.abs(_np(Z) - Zo).max() <= 1e-5
    np.testing.assert_array_equal(_np(M).astype(bool), Mo)
    rgb, idx = taxim.shade(Z, return_bins=True)
    idx = _np(idx).astype(np.int64)
    same = (idx[..., 0] == imo) & (idx[..., 1] == ido)
    flat = mago == 0
    assert same[flat].all(), "flat pixels must land in bin (0, 62) exactly like the deterministic oracle"
    strong = mago > 1e-3
    assert same[strong].mean() >= 0.99
    err = rgb_rel_err(_np(rgb), rgbo)
    assert err[same].max() <= 1e-4
    print(shape, "bin-equal frac", same.mean(), "rgb rel err same-bin", err[same].max())


def test_indentation_depth_kernel(calib_dir):
    from oracle.taxim_oracle import TaximOracle
    from tacex_amd import _lib
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    hm, _ = synthetic_depth_maps(9, 240, 320, seed=5, flat_fraction=0.3)
    hm[3] = 23.5  # object closer than the sensor case -> clamps to gelpad height (TS:124)
    hm[4] = 28.5  # exactly touching
    lib = _lib.load_library()
    d = hm.cuda()
    fmin = torch.empty(9, device="cuda")
    ind = torch.empty(9, device="cuda")
    _lib.check(lib.tacex_indentation_depth(d.data_ptr(), 0.0045, 0.024, fmin.data_ptr(), ind.data_ptr(), 0, 9, 240, 320,
                                           torch.cuda.current_stream().cuda_stream), "indent")
    np.testing.assert_array_equal(_np(fmin), hm.numpy().min(axis=(1, 2)))
    np.testing.assert_array_equal(_np(ind), TaximOracle.indentation_depth(hm.numpy()))
    # the same pass with the contact row range as a by-product (row-wise kernel): identical minimum / indentation, and
    # rows[:, 0:2] = first / last row with S = (hm - min) - indent < 0 in float32 (TT:441), (H, -1) for frames without contact
    fmin2, ind2 = torch.empty(9, device="cuda"), torch.empty(9, device="cuda")
    rows = torch.full((9, 4), 77, dtype=torch.int32, device="cuda")
    _lib.check(lib.tacex_indentation_depth(d.data_ptr(), 0.0045, 0.024, fmin2.data_ptr(), ind2.data_ptr(), rows.data_ptr(), 9, 240, 320,
                                           torch.cuda.current_stream().cuda_stream), "indent+rows")
    np.testing.assert_array_equal(_np(fmin2), _np(fmin))
    np.testing.assert_array_equal(_np(ind2), _np(ind))
    S = (hm - fmin.cpu().view(-1, 1, 1)) - ind.cpu().view(-1, 1, 1)
    has = (S < 0).any(2).numpy()
    hasc = (S < 0).any(1).numpy()  # ... and the first / last such COLUMN, (W, -1) without contact (zero-block skipping of the band levels)
    want = np.array([[np.where(r)[0][0], np.where(r)[0][-1], np.where(c)[0][0], np.where(c)[0][-1]] if r.any() else [240, -1, 320, -1]
                     for r, c in zip(has, hasc)], dtype=np.int32)
    np.testing.assert_array_equal(_np(rows), want)
    assert (want[:, 1] >= 0).any() and (want[:, 1] < 0).any() and (want[:, 3] - want[:, 2] < 200).any()


def test_zero_band_skipping_is_exact(calib_dir, tmp_path):
    """Bands of the band levels whose input window lies outside the contact rows are stored as zeros without being computed
    (frame_rows_kernel + blur_mfma_kernel).  The result must not differ in a single bit from the full computation
    (TACEX_BAND_SKIP=0, read once per process): small contacts, contact at the top / bottom border (reflect padding), frames
    without contact, through render_direct (library-side rows) and through the sensor (rows from the depth pass)."""
    import os
    import subprocess
    import sys

    from conftest import REPO

    script = tmp_path / "skip_run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "from tacex_amd import GelSightSensor, GelSightSensorCfg\n"
        "from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg\n"
        "from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim\n"
        "from tacex_amd.utils.synthetic import synthetic_depth_maps\n"
        "H, W, n = 240, 320, 12\n"
        "hm, ind = synthetic_depth_maps(n, H, W, seed=3, flat_fraction=0.25)\n"
        "yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')\n"
        "for b, (cy, cx, r) in enumerate([(6, 40, 9.0), (H - 4, 200, 7.0), (120, 160, 5.0), (60, 300, 12.0)]):\n"
        "    d2 = ((yy - cy) ** 2 + (xx - cx) ** 2).float()\n"
        "    hm[b] = torch.where(d2 < r * r, 28.0 + 0.02 * d2.sqrt(), torch.full_like(d2, 29.0))\n"
        f"t = Taxim(calib_folder={str(calib_dir)!r}, backend='hip', device='cuda:0')\n"
        "Z, M = t.deform(hm.cuda(), torch.full((n,), 0.8).cuda())\n"
        "rgb = t.render_direct(hm.cuda(), with_shadow=False, press_depth=torch.full((n,), 0.8).cuda())\n"
        "cfg = GelSightSensorCfg(num_envs=n, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),\n"
        "    data_types=['tactile_rgb', 'height_map'],\n"
        f"    optical_sim_cfg=TaximSimulatorCfg(calib_folder_path={str(calib_dir)!r}, gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,\n"
        "        tactile_img_res=(W, H), device='cuda:0'), marker_motion_sim_cfg=None, device='cuda:0')\n"
        "s = GelSightSensor(cfg); s.initialize()\n"
        "s.set_camera_depth((hm / 1000.0).cuda())\n"
        "s.update(0.01, force_recompute=True)\n"
        "rows = s.optical_simulator._frame_rows.cpu().numpy()\n"
        "np.savez(sys.argv[1], Z=Z.cpu().numpy(), M=M.cpu().numpy(), rgb=rgb.cpu().numpy(), srgb=s.data.output['tactile_rgb'].cpu().numpy(), rows=rows)\n")
    outs = {}
    for skip in ("1", "0"):
        out = tmp_path / f"s{skip}.npz"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, TACEX_BAND_SKIP=skip), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[skip] = np.load(out)
    a, b = outs["1"], outs["0"]
    for k in ("Z", "M", "rgb", "srgb"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    rows = a["rows"]
    assert (rows[:, 1] < 0).any(), "a frame without contact"
    assert ((rows[:, 1] >= 0) & (rows[:, 1] - rows[:, 0] < 40)).any(), "a small contact (most bands skipped)"
    assert (rows[:, 0] == 0).any() and (rows[:, 1] == 239).any(), "contacts on the top and bottom border"
    assert np.abs(a["Z"]).max() > 0.1


def test_band_levels_on_two_streams_change_no_bit(calib_dir, tmp_path):
    """The chunks of a pass alternate between the caller's stream and a stream of the context's own (TACEX_LEVEL_STREAMS, default 2,
    read once per process; `pipeline_impl`): fork / join events order them against the pass's inputs and its tail.  A 768-frame pass
    (three chunks per level at one stream, six at two) rendered with one and with two streams - twice in a row on the same context,
    so that the second pass's fork meets the first one's tail - must agree bit for bit: frames (SHA-256 of the whole batch) and the
    fused observation."""
    import hashlib
    import os
    import subprocess
    import sys

    from conftest import REPO

    script = tmp_path / "lvl_run.py"
    script.write_text(
        "import sys, hashlib, numpy as np, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim\n"
        "from tacex_amd.utils.synthetic import synthetic_depth_maps\n"
        "B, H, W = 768, 240, 320\n"
        f"t = Taxim(calib_folder={str(calib_dir)!r}, backend='hip', device='cuda:0')\n"
        "out = torch.empty((B, H, W, 3), device='cuda:0')\n"
        "obs = torch.empty((B, 32, 32, 3), dtype=torch.uint8, device='cuda:0')\n"
        "hs = []\n"
        "for seed in (1, 2):\n"
        "    hm, ind = synthetic_depth_maps(B, H, W, seed=seed, device='cuda:0')\n"
        "    t.render_direct(hm, False, ind, out=out, obs_out=obs)\n"
        "    torch.cuda.synchronize()\n"
        "    hs.append(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())\n"
        "    hs.append(hashlib.sha256(obs.cpu().numpy().tobytes()).hexdigest())\n"
        "    hs.append(str(float(out.abs().sum())))\n"
        "open(sys.argv[1], 'w').write('\\n'.join(hs))\n")
    outs = {}
    for n in ("1", "2"):
        out = tmp_path / f"l{n}.txt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, TACEX_LEVEL_STREAMS=n), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[n] = out.read_text().split("\n")
    assert outs["1"] == outs["2"], (outs["1"], outs["2"])
    assert float(outs["2"][2]) > 0.0 and outs["2"][0] != outs["2"][3]  # real frames, and the two batches differ


def test_no_shift_render_and_numpy_entry(taxim, calib_dir):
    """press_depth=None renders the height map as is (TT:188-189 skipped); render() takes NumPy (TT:166-171)."""
    from oracle.taxim_oracle import TaximOracle

    o = TaximOracle(calib_dir, (48, 64), "direct")
    yy, xx = np.meshgrid(np.arange(48.0), np.arange(64.0), indexing="ij")
    S = (0.5 - np.sqrt(np.clip(400 - (xx - 30) ** 2 - (yy - 20) ** 2, 0, None)) * 0.05).astype(np.float32)[None]
    S = np.minimum(S, 0.6)
    out = taxim.render(S, with_shadow=False, press_depth=None)
    assert out.shape == (1, 48, 64, 3)
    assert_same_bin_vs_oracle(taxim, o, S, None, out)


def test_resize_kernel_matches_torch():
    from tacex_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(0)
    for (sh, sw, dh, dw) in [(240, 320, 32, 32), (24, 32, 240, 320), (240, 320, 480, 640), (100, 77, 33, 50)]:
        x = torch.rand((3, sh, sw), generator=g).cuda()
        # smooth it a little so fp32 weight roundoff in torch's own kernel stays small
        y = torch.empty((3, dh, dw), device="cuda")
        _lib.check(lib.tacex_resize_bilinear_aa(x.data_ptr(), sh, sw, y.data_ptr(), dh, dw, 3,
                                                torch.cuda.current_stream().cuda_stream), "resize")
        ref = torch.nn.functional.interpolate(x.cpu()[None], size=[dh, dw], mode="bilinear", antialias=True)[0]
        assert (y.cpu() - ref).abs().max() < 2e-4, (sh, sw, dh, dw)


def test_errors_are_loud(taxim):
    with pytest.raises(ValueError):
        taxim.render_direct(torch.zeros((1, 32, 32)), with_shadow=False)  # CPU tensor
    from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim

    with pytest.raises(ValueError):
        Taxim(backend="nope")
    with pytest.raises(ImportError):
        Taxim(backend="jax")


@pytest.mark.parametrize("shape", [(240, 320), (480, 640)])
def test_fused_tail_matches_unfused_levels(taxim, shape):
    """The fused tail kernel (levels k=9,5,3,5 + shade in LDS tiles) must reproduce the level-by-level kernels:
    same contact mask, deformed gel to float32 roundoff (different but equivalent summation order), same RGB on
    same-bin pixels - including border tiles (reflect padding per level) and the partial last tile row."""
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    H, W = shape
    hm, ind = synthetic_depth_maps(5, H, W, seed=321, flat_fraction=0.2)
    # put one contact right at the image corner / border to exercise the mirrored halo of border tiles
    hm[0, : H // 6, : W // 6] = torch.minimum(hm[0, : H // 6, : W // 6], torch.tensor(28.0))
    hm[1, -H // 8 :, W // 3 : W // 2] = 27.9
    hm, ind = hm.cuda(), None
    from oracle.taxim_oracle import TaximOracle

    indent = torch.from_numpy(TaximOracle.indentation_depth(hm.cpu().numpy())).cuda()
    res = {}
    for fused in (True, False):
        taxim.set_fused_tail(shape, fused)
        z = torch.empty_like(hm)
        m = torch.empty(hm.shape, dtype=torch.uint8, device="cuda")
        rgb = taxim.render_direct(hm, False, indent, z_out=z, mask_out=m).movedim(1, 3).clone()
        _, idx = taxim.shade(z, return_bins=True)
        res[fused] = (z.clone(), m.clone(), rgb, idx.clone())
    taxim.set_fused_tail(shape, True)
    zf, mf, rf, idf = res[True]
    zu, mu, ru, idu = res[False]
    assert torch.equal(mf, mu)
    assert (zf - zu).abs().max().item() <= 2e-6
    same = (idf == idu).all(-1)
    assert same.float().mean().item() > 0.995
    assert ((rf - ru).abs()[same]).max().item() <= 2e-6


def test_policy_observation_downsample_matches_torch(taxim):
    """(B,240,320,3) -> (B,32,32,3) antialiased bilinear (two-pass kernel) vs torch's own antialiased interpolate."""
    from tacex_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(3)
    x = torch.rand((4, 240, 320, 3), generator=g)
    x = torch.cumsum(x, 2) / 160.0  # smooth along x so torch's float32 weight roundoff stays small
    xd = x.cuda()
    y = torch.empty((4, 32, 32, 3), device="cuda")
    tmp = torch.empty((4, 32, 320, 3), device="cuda")
    for t in (tmp, None):
        y.zero_()
        _lib.check(lib.tacex_resize_bilinear_aa_nhwc(xd.data_ptr(), 240, 320, y.data_ptr(), 32, 32, 3, 4,
                                                     0 if t is None else t.data_ptr(), torch.cuda.current_stream().cuda_stream), "resize")
        ref = torch.nn.functional.interpolate(x.movedim(3, 1), size=[32, 32], mode="bilinear", antialias=True).movedim(1, 3)
        assert (y.cpu() - ref).abs().max() < 1e-5


def test_shadow_branch_vs_reference_and_oracle(taxim, golden_dir, calib_dir):
    """with_shadow=True (TT:260-346): ring detection, 4x51 ray march with float atomic-min, two image blurs.
    vs the reference on EVERY pixel whose 7x7 receptive field is well conditioned, and vs the oracle on every pixel whose
    field holds no bin flip between the two (tests/studies/shadow_outliers.py attributes the rest)."""
    from oracle.taxim_oracle import TaximOracle
    from parity import well_conditioned_field

    g = dict(np.load(golden_dir / "taxim_240x320.npz"))
    hm = torch.from_numpy(g["hm"]).cuda()
    indent = torch.from_numpy(g["indent"]).cuda()
    rgb = taxim.render_direct(hm, with_shadow=True, press_depth=indent).movedim(1, 3).cpu().numpy()
    assert rgb.shape == g["rgb_shadow"].shape and rgb.min() >= 0 and rgb.max() <= 1
    Z, M = taxim.deform(hm, indent)
    _, idx = taxim.shade(Z, return_bins=True)
    idx = idx.cpu().numpy().astype(np.int64)
    ok = well_conditioned_field(idx[..., 0], idx[..., 1], g)
    assert ok.sum() > 50000
    # EVERY such pixel within the 1e-4 relative tolerance (measured 2.5e-6; tests/studies/shadow_outliers.py: on this fixture no
    # shadow sample of the HIP path lands on another pixel than the reference's)
    from parity import rgb_rel_err
    d = rgb_rel_err(rgb, g["rgb_shadow"])
    assert d[ok].max() <= 1e-4, d[ok].max()
    assert np.abs(g["rgb_shadow"] - g["rgb"])[ok].max() > 0.1  # shadows are really cast there
    o = TaximOracle(calib_dir, (240, 320), "direct")
    ref = o.render_direct(g["hm"], g["indent"], with_shadow=True)
    do = np.abs(rgb - ref)
    # vs the oracle: identical to float32 round-off wherever the 7x7 field of the two blurs holds no bin flip (0.25 % of the
    # pixels flip: flat gel, where the direction is round-off of either atan2); the flips themselves are bounded below
    from scipy import ndimage
    Zo, _ = o.gel_pad_deformation(o.shifted_height_map(g["hm"], g["indent"]))
    im_o, id_o = o.bins(*o.normals(-(Zo / np.float32(o.p.pixmm))))
    same = (idx[..., 0] == im_o) & (idx[..., 1] == id_o)
    field = np.stack([ndimage.binary_erosion(same[b], structure=np.ones((7, 7)), border_value=1) for b in range(same.shape[0])])
    assert field.mean() > 0.95, field.mean()
    assert do[field].max() <= 1e-6, do[field].max()
    assert (do > 1e-3).mean() < 5e-3
    # no-contact frame: no ring -> both blurs of (flat shade + background) only
    assert do[-1].max() <= 1e-5


def test_mfma_and_valu_band_kernels_agree(calib_dir, golden_dir, tmp_path):
    """The matrix-core (v_mfma_f32_16x16x4_f32, default) and the VALU band kernels (TACEX_BLUR_MFMA=0) must give the same
    deformation.  The switch is read once per process: child processes."""
    import subprocess
    import sys

    from conftest import REPO

    script = tmp_path / "mfma_run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim\n"
        f"g = np.load({str(golden_dir / 'taxim_240x320.npz')!r})\n"
        f"t = Taxim(calib_folder={str(calib_dir)!r}, backend='hip', device='cuda:0')\n"
        "Z, M = t.deform(torch.from_numpy(g['hm']).cuda(), torch.from_numpy(g['indent']).cuda())\n"
        "np.save(sys.argv[1], Z.cpu().numpy()); np.save(sys.argv[1] + '.m.npy', M.cpu().numpy())\n")
    outs = {}
    for flag, tiles in (("0", "1"), ("1", "1")):
        out = tmp_path / f"z{flag}{tiles}.npy"
        env = dict(__import__("os").environ, TACEX_BLUR_MFMA=flag, TACEX_MFMA_TILES=tiles)
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag + tiles] = (np.load(out), np.load(str(out) + ".m.npy"))
    g = np.load(golden_dir / "taxim_240x320.npz")
    for key in ("11",):
        np.testing.assert_array_equal(outs[key][1], outs["01"][1])
        assert np.abs(outs[key][0] - outs["01"][0]).max() <= 2e-6
        assert np.abs(outs[key][0] - g["Z"]).max() <= 1e-5


@pytest.mark.parametrize("shape,segs,mcols", [((240, 320), "0", 11), ((240, 320), "3", 11), ((240, 320), "8", 11), ((480, 640), "0", 11),
                                              ((480, 640), "5", 11), ((240, 320), "0", 26)])
def test_streaming_tail_matches_tiled_tail(calib_dir, tmp_path, shape, segs, mcols):
    """The wave-autonomous streaming tail (taxim_stream.hip, default) against the LDS-tiled tail (taxim_tail.hip) through the
    sensor boundary: RGB bit-equal (same summation order of every level, same shading code), uint8 / float policy observation
    to round-off (different partial-sum tiling), FOTS markers bit-equal (marker-pixel values and integer contact statistics
    are the same numbers).  Vertical segmentation (TACEX_STREAM_SEGS, read once per process) must not change a bit of RGB:
    strips / segments only differ in how much warm-up they recompute.  mcols = 26 puts more markers on a row than the packed
    marker slots of the row table hold (the kernel then walks the marker CSR)."""
    import os
    import subprocess
    import sys

    from conftest import REPO

    H, W = shape
    mrows = 9 if mcols == 11 else 4  # the FOTS kernels take up to 128 markers
    script = tmp_path / "stream_run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "from tacex_amd import GelSightSensor, GelSightSensorCfg\n"
        "from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg\n"
        "from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg\n"
        "from tacex_amd.utils.synthetic import synthetic_depth_maps\n"
        f"H, W, n = {H}, {W}, 6\n"
        "mode = int(sys.argv[2])\n"
        "res = {}\n"
        "for dt in ('uint8', 'float32'):\n"
        "    cfg = GelSightSensorCfg(num_envs=n, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),\n"
        "        data_types=['tactile_rgb', 'height_map', 'marker_motion'],\n"
        f"        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path={str(calib_dir)!r}, gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,\n"
        "            tactile_img_res=(W, H), device='cuda:0', policy_obs_res=(32, 32), policy_obs_dtype=dt),\n"
        "        marker_motion_sim_cfg=FOTSMarkerSimulatorCfg(tactile_img_res=(W, H), device='cuda:0',\n"
        f"            marker_params=FOTSMarkerSimulatorCfg.MarkerParams(num_markers_col={mcols}, num_markers_row={mrows}, num_markers={mcols * mrows}, x0=15 * W // 320, y0=26 * H // 240,\n"
        f"                dx={26.0 * 11 / mcols}, dy=29.0)), device='cuda:0')\n"
        "    s = GelSightSensor(cfg); s.initialize()\n"
        "    s.optical_simulator._taxim.set_fused_tail((H, W), mode)\n"
        "    hm, _ = synthetic_depth_maps(n, H, W, seed=99, flat_fraction=0.17)\n"
        "    hm[0, : H // 6, : W // 6] = torch.minimum(hm[0, : H // 6, : W // 6], torch.tensor(28.0))   # contact in the image corner\n"
        "    hm[1, -H // 8 :, W // 3 : W // 2] = 27.9                                                    # ... and on the bottom border\n"
        "    hm[2, H // 2 - 3 : H // 2 + 3, W // 2 - 40 : W // 2 + 40] = 27.5                            # across the strip seam\n"
        "    s.set_camera_depth((hm / 1000.0).cuda())\n"
        "    for k in range(2):\n"
        "        s.marker_motion_simulator.set_indenter_yaw(torch.full((n,), 0.05 * k, device='cuda:0'))\n"
        "        s.update(0.01, force_recompute=True)\n"
        "    o = s.data.output\n"
        "    res['rgb'] = o['tactile_rgb'].cpu().numpy(); res['obs_' + dt] = o['tactile_rgb_obs'].cpu().numpy()\n"
        "    res['markers'] = o['marker_motion'].cpu().numpy(); res['pix_z'] = s.optical_simulator._pix_z.cpu().numpy()\n"
        "    res['pix_m'] = s.optical_simulator._pix_m.cpu().numpy(); res['traj'] = s.marker_motion_simulator._traj_state.cpu().numpy()\n"
        "np.savez(sys.argv[1], **res)\n")
    outs = {}
    for mode in ("1", "2"):
        out = tmp_path / f"m{mode}.npz"
        env = dict(os.environ, TACEX_STREAM_SEGS=segs)
        r = subprocess.run([sys.executable, str(script), str(out), mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = np.load(out)
    a, b = outs["1"], outs["2"]
    np.testing.assert_array_equal(a["rgb"], b["rgb"])
    np.testing.assert_array_equal(a["pix_m"], b["pix_m"])
    np.testing.assert_array_equal(a["pix_z"], b["pix_z"])
    np.testing.assert_array_equal(a["traj"], b["traj"])
    np.testing.assert_array_equal(a["markers"], b["markers"])
    assert np.abs(a["obs_float32"] - b["obs_float32"]).max() <= 2e-6
    assert np.abs(a["obs_uint8"].astype(int) - b["obs_uint8"].astype(int)).max() <= 1 and (a["obs_uint8"] == b["obs_uint8"]).mean() > 0.999
    assert a["pix_m"].sum() > 0 and np.abs(a["markers"][:, 1] - a["markers"][:, 0]).max() > 0.1


@pytest.mark.parametrize("shape", [(240, 320), (480, 640)])
def test_shadow_ray_samples_exact_vs_oracle(taxim, golden_dir, calib_dir, shape):
    """The ray march of the shadow branch is integer work (ring pixels, direction / height bins, truncated sample pixels,
    TT:261-337) fed by float32 expressions in the reference's op order: given the SAME deformed gel, contact mask and gradient
    direction, the HIP kernel's per-pixel / channel minimum map must equal the oracle's EXACTLY - every sample pixel and value."""
    from oracle.taxim_oracle import TaximOracle
    from parity import unpack_mask

    H, W = shape
    g = dict(np.load(golden_dir / f"taxim_{H}x{W}.npz"))
    n = 2 if H == 480 else 4
    Z = g["Z"][:n].astype(np.float32)
    M = unpack_mask(g["M"], g["Z"].shape)[:n]
    o = TaximOracle(calib_dir, (H, W), "direct")
    ref, gdir = o.shadow_map(Z, M)
    assert np.isfinite(ref).any(), "no shadow samples at all"
    got = taxim.shadow_rays(torch.from_numpy(Z).cuda(), torch.from_numpy(M.astype(np.uint8)).cuda(), torch.from_numpy(gdir.astype(np.float32)).cuda())
    got = got.cpu().numpy()
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(ref))  # the sample index set
    np.testing.assert_array_equal(got, ref)                            # ... and the table values that landed there


def test_shadow_branch_640x480_vs_reference(taxim, golden_dir):
    """with_shadow=True at BASELINE config C5's resolution against the reference's own render (frame 0 of the 480x640 fixture):
    same protocol as at 320x240 - ALL pixels whose receptive field of the two blurs (k = 5 and k = 9 here) is well conditioned."""
    from parity import well_conditioned_field

    g = dict(np.load(golden_dir / "taxim_480x640.npz"))
    hm = torch.from_numpy(g["hm"][:1]).cuda()
    indent = torch.from_numpy(g["indent"][:1]).cuda()
    rgb = taxim.render_direct(hm, with_shadow=True, press_depth=indent).movedim(1, 3).cpu().numpy()
    ref = g["rgb_shadow"]
    assert rgb.shape == ref.shape == (1, 480, 640, 3)
    Z, _ = taxim.deform(hm, indent)
    _, idx = taxim.shade(Z, return_bins=True)
    idx = idx.cpu().numpy().astype(np.int64)
    gg = {"idx_mag": g["idx_mag"], "idx_dir": g["idx_dir"], "grad_mag": np.where(g["idx_mag"] > 0, 1.0, 0.0)}  # slim fixture: no grad_mag
    ok = well_conditioned_field(idx[..., 0], idx[..., 1], gg, frames=slice(0, 1), radius=6)
    assert ok.sum() > 20000
    from parity import rgb_rel_err
    d = rgb_rel_err(rgb, ref)
    assert d[ok].max() <= 1e-4, d[ok].max()  # every such pixel (measured 2.9e-6)
    assert np.abs(ref - g["rgb"][:1])[ok].max() > 0.05  # shadows are really cast there
