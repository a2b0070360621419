"""CPU: pin the oracle (oracle/*.py) against golden vectors produced by the reference itself."""
import numpy as np
import pytest

from oracle.fots_oracle import FOTSOracle, MarkerMotionOracle
from oracle.taxim_oracle import TaximOracle, gaussian_kernel1d, gaussian_kernel_size
from parity import check_against_reference, unpack_mask

SHAPES = [(32, 32), (24, 32), (48, 64), (240, 320), (480, 640)]


def _load(golden_dir, H, W):
    return dict(np.load(golden_dir / f"taxim_{H}x{W}.npz"))


def test_gaussian_kernel_tables(golden_dir):
    tb = np.load(golden_dir / "taxim_tables.npz")
    for (H, W) in [(240, 320), (480, 640), (32, 32), (24, 32), (48, 64)]:
        sig, ks = tb[f"sigma_{H}x{W}"], tb[f"ksize_{H}x{W}"]
        for li in range(7):
            kw, kh = gaussian_kernel_size(sig[li, 0]), gaussian_kernel_size(sig[li, 1])
            assert (kw, kh) == tuple(ks[li]), (H, W, li)
            np.testing.assert_allclose(gaussian_kernel1d(sig[li, 0], kw), tb[f"taps_w_{H}x{W}_{li}"], rtol=0, atol=3e-8)
            np.testing.assert_allclose(gaussian_kernel1d(sig[li, 1], kh), tb[f"taps_h_{H}x{W}_{li}"], rtol=0, atol=3e-8)
    # SURVEY.md Appendix B.1
    assert tb["ksize_240x320"][:, 0].tolist() == [61, 33, 17, 9, 5, 3, 5]
    assert tb["ksize_480x640"][:, 0].tolist() == [117, 61, 33, 15, 9, 5, 9]


@pytest.mark.parametrize("shape", SHAPES)
def test_tables_match_reference(golden_dir, calib_dir, shape):
    H, W = shape
    g = _load(golden_dir, H, W)
    o = TaximOracle(calib_dir, shape, "direct")
    assert np.abs(o.gel - g["gel"]).max() <= 1e-6
    if "bg" in g:
        assert np.abs(o.bg - g["bg"]).max() <= 2e-6


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("blur", ["direct", "fft32"])
def test_taxim_oracle_vs_reference(golden_dir, calib_dir, shape, blur):
    H, W = shape
    if blur == "fft32" and H >= 480:
        pytest.skip("fft32 mode is exercised at the smaller sizes")
    g = _load(golden_dir, H, W)
    o = TaximOracle(calib_dir, shape, blur)
    hm, indent = g["hm"], g["indent"]
    np.testing.assert_array_equal(o.indentation_depth(hm), indent)
    S = o.shifted_height_map(hm, indent)
    if "S" in g:
        np.testing.assert_array_equal(S, g["S"])
    if "Z_levels" in g:
        nf = g["Z_levels"].shape[1]
        Z, M, J, levels = o.gel_pad_deformation(S, return_levels=True)
        assert np.abs(J - g["J"]).max() <= 1e-6  # the reference gel map carries ~4e-7 FFT roundoff
        for li, zl in enumerate(levels):
            assert np.abs(zl[:nf] - g["Z_levels"][li]).max() <= 1e-5, li
    else:
        Z, M = o.gel_pad_deformation(S)
    np.testing.assert_array_equal(M, unpack_mask(g["M"], M.shape))
    rgb, mag, dr, im, idd = o.shade(Z, return_all=True)
    stats = check_against_reference(Z, im, idd, rgb, g)
    print(shape, blur, stats)
    # end-to-end entry agrees with the staged path
    np.testing.assert_array_equal(o.render_direct(hm, indent), rgb)


def test_flat_frames_are_background(golden_dir, calib_dir):
    """No-contact frames: the deterministic oracle returns exactly bg + poly(bin 0, 62); the reference's
    own output differs there only through its arbitrary direction bins."""
    g = _load(golden_dir, 240, 320)
    o = TaximOracle(calib_dir, (240, 320), "direct")
    rgb, mag, dr, im, idd = o.shade(o.gel_pad_deformation(o.shifted_height_map(g["hm"][-1:], g["indent"][-1:]))[0], True)
    assert (im == 0).all() and (idd == 62).all()
    assert np.abs(rgb - g["rgb"][-1:]).max() < 0.13  # bounded by the dir-bin spread at mag-bin 0


def test_marker_grid_bit_exact(golden_dir):
    g = np.load(golden_dir / "fots_240x320.npz")
    mm = MarkerMotionOracle()
    np.testing.assert_array_equal(mm.init_marker_pos(), g["init_marker_pos"])
    assert mm.init_x[0].tolist() == [15, 44, 73, 102, 131, 160, 189, 218, 247, 276, 305]
    assert mm.init_y[:, 0].tolist() == [26, 49, 73, 96, 120, 143, 167, 190, 214]


def test_fots_oracle_vs_reference(golden_dir, calib_dir):
    g = np.load(golden_dir / "fots_240x320.npz")
    o = TaximOracle(calib_dir, (240, 320), "direct")
    steps, n = g["hm"].shape[:2]
    fo = FOTSOracle(o, n)
    for s in range(steps):
        md = fo.step(g["hm"][s], g["indent"][s], g["theta"][s])
        ref = g["marker_data"][s]
        np.testing.assert_array_equal(md[:, 0], ref[:, 0])
        # float32 output of float64 arithmetic on a deformed gel that differs by ~1e-6 mm
        assert np.abs(md[:, 1] - ref[:, 1]).max() <= 1e-4, s
        disp = np.abs(ref[:, 1] - ref[:, 0]).max()
        assert disp > 0.1  # the fixture really moves markers


def test_torch_cpu_port_vs_reference(golden_dir, calib_dir):
    """The bench's CPU-baseline port reproduces the reference RGB on same-bin pixels (it is the reference's own
    algorithm incl. the FFT blur, so flat-region noise differs only through FFT library roundoff)."""
    import torch

    from oracle.taxim_torch_cpu import TaximTorchCpuPort

    g = _load(golden_dir, 240, 320)
    port = TaximTorchCpuPort(calib_dir, (240, 320))
    rgb = port.render_direct(torch.from_numpy(g["hm"]), torch.from_numpy(g["indent"])).numpy()
    assert rgb.shape == g["rgb"].shape
    strong = g["grad_mag"] > 1e-3
    d = np.abs(rgb - g["rgb"])
    assert np.quantile(d[strong], 0.99) <= 1e-4
    assert (d[strong] <= 1e-4).mean() >= 0.99


def test_shadow_branch_oracle_vs_reference(golden_dir, calib_dir):
    """with_shadow=True (TT:260-346): boundary ring, 4-ray fan x 51 steps, scatter-min, two image blurs."""
    from parity import well_conditioned_field

    g = _load(golden_dir, 240, 320)
    o = TaximOracle(calib_dir, (240, 320), "direct")
    S = o.shifted_height_map(g["hm"], g["indent"])
    Z, M = o.gel_pad_deformation(S)
    _, mag, dr, im, idd = o.shade(Z, True)
    rgb = o.shade_with_shadow(Z, M)
    ok = well_conditioned_field(im, idd, g)
    assert ok.sum() > 50000
    d = np.abs(rgb - g["rgb_shadow"])
    assert d[ok].max() <= 1e-5
    # the fixture really casts shadows on comparable pixels
    assert np.abs(g["rgb_shadow"] - g["rgb"])[ok].max() > 0.1
    assert o.shadow_attachment_rounds()[0].tolist() == [2, 2] and o.shadow_attachment_rounds()[1].tolist() == [3, 3]
    # small images: blur kernels degenerate to k=1, no boundary pixel casts a shadow -> identical to the plain path
    g48 = _load(golden_dir, 48, 64)
    o48 = TaximOracle(calib_dir, (48, 64), "direct")
    r48 = o48.render_direct(g48["hm"], g48["indent"], with_shadow=True)
    strong = g48["grad_mag"] > 1e-3
    assert np.quantile(np.abs(r48 - g48["rgb_shadow"])[strong], 0.99) <= 1e-4


def test_shadow_branch_oracle_vs_reference_640x480(golden_dir, calib_dir):
    """The oracle's shadow branch at BASELINE config C5's resolution against the reference's render (frame 0)."""
    from parity import well_conditioned_field

    g = _load(golden_dir, 480, 640)
    o = TaximOracle(calib_dir, (480, 640), "direct")
    hm, ind = g["hm"][:1], g["indent"][:1]
    Z, M = o.gel_pad_deformation(o.shifted_height_map(hm, ind))
    _, mag, dr, im, idd = o.shade(Z, True)
    rgb = o.shade_with_shadow(Z, M)
    gg = {"idx_mag": g["idx_mag"], "idx_dir": g["idx_dir"], "grad_mag": np.where(g["idx_mag"] > 0, 1.0, 0.0)}  # slim fixture
    ok = well_conditioned_field(im, idd, gg, frames=slice(0, 1), radius=6)
    assert ok.sum() > 20000
    d = np.abs(rgb - g["rgb_shadow"])
    # a handful of the ~10^5 ray samples land in the neighbouring pixel (NumPy's and torch's float32 cos / sin differ in the last
    # bit for some fan angles, and the product is truncated to a pixel index); the two blurs spread each over ~100 pixels
    assert np.quantile(d[ok], 0.999) <= 1e-4, np.quantile(d[ok], 0.999)
    assert (d[ok] > 1e-3).mean() == 0.0 and d[ok].max() <= 5e-4
    assert np.abs(g["rgb_shadow"] - g["rgb"][:1])[ok].max() > 0.05
