"""Parity protocol helpers shared by CPU and GPU tests ('Parity protocol').

The reference's RGB is ill-conditioned where the gel is flat (gradient = FFT roundoff, direction bin
arbitrary), so comparisons against *reference* outputs are restricted to well-conditioned pixels:
  (1) deformed gel   : max |dZ| <= 1e-5 mm
  (2) bin indices    : equal on >= 99 % of pixels with grad_mag > 1e-3
  (3) RGB            : <= 1e-4 relative, |d| <= 1e-4 * max(|ref|, 0.05) (RGB lives in [0,1]; the 0.05 floor keeps near-black
                       channels from turning float32 round-off into a relative error), on same-bin pixels
Comparisons between the HIP path and the deterministic oracle use every pixel (flat regions included).
"""
import numpy as np


def unpack_mask(packed, shape):
    n = int(np.prod(shape))
    return np.unpackbits(packed)[:n].reshape(shape).astype(bool)


def rgb_rel_err(a, b):
    """1e-4 *relative* RGB tolerance (north_star): |a-b| / max(|b|, 0.05) - RGB lives in [0,1]."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 0.05)


def check_against_reference(Z, im, idd, rgb, g, frames=None, z_tol=1e-5, rgb_tol=1e-4, min_bin_frac=0.99):
    """g: golden dict (reference outputs). Arrays are (B,H,W[,3]). Returns a dict of measured stats."""
    sl = slice(None) if frames is None else frames
    Zr = g["Z"][sl]
    stats = {"z_maxdiff": float(np.abs(np.asarray(Z, np.float64) - Zr).max())}
    assert stats["z_maxdiff"] <= z_tol, stats
    imr = g["idx_mag"][sl].astype(np.int64)
    idr = g["idx_dir"][sl].astype(np.int64)
    strong = g["grad_mag"][sl] > 1e-3 if "grad_mag" in g else imr > 0
    same = (np.asarray(im) == imr) & (np.asarray(idd) == idr)
    if strong.any():
        stats["bin_equal_frac_strong"] = float(same[strong].mean())
        assert stats["bin_equal_frac_strong"] >= min_bin_frac, stats
    stats["bin_equal_frac_all"] = float(same.mean())
    err = rgb_rel_err(rgb, g["rgb"][sl])
    sb = same & strong
    if sb.any():
        stats["rgb_rel_same_bin_strong"] = float(err[sb].max())
        assert stats["rgb_rel_same_bin_strong"] <= rgb_tol, stats
    if same.any():
        stats["rgb_rel_same_bin_all"] = float(err[same].max())
        assert stats["rgb_rel_same_bin_all"] <= rgb_tol, stats
    return stats


def well_conditioned_field(im, idd, g, frames=None, radius=3):
    """Pixels whose whole (2r+1)^2 receptive field is same-bin AND strong-gradient w.r.t. the reference: the shadow
    branch blurs the shaded image twice (k=3 then k=5 at 320x240), which smears the reference's arbitrary flat-region
    bins over their neighbourhood, so only such pixels are comparable with the reference's shadow output."""
    from scipy import ndimage

    sl = slice(None) if frames is None else frames
    same = (np.asarray(im) == g["idx_mag"][sl]) & (np.asarray(idd) == g["idx_dir"][sl]) & (g["grad_mag"][sl] > 1e-3)
    st = np.ones((2 * radius + 1, 2 * radius + 1))
    return np.stack([ndimage.binary_erosion(same[b], structure=st) for b in range(same.shape[0])])


def assert_same_bin_vs_oracle(taxim, oracle, hm, press, out_nhwc, rgb_tol=1e-4, min_bin_frac=0.99, z_tol=1e-5):
    """HIP render vs the deterministic oracle by the same-bin protocol, with the MAXIMUM (no quantile): the HIP path's own deformed
    gel (`taxim.deform` on the same input; press None = the no-shift entry) within z_tol of the oracle's, its bins (`taxim.shade`
    with bins) equal to the oracle's on >= 99 % of the strong-gradient pixels, and every same-bin pixel of the rendered RGB within
    1e-4 relative.  Pixels whose bin differs sit on a bin edge of one of the two float paths: a neighbouring table record, not an
    arithmetic error - they are counted, not compared."""
    import torch

    hm = np.asarray(hm, np.float32)
    if press is None:
        S = hm
        Z, _ = taxim.deform(torch.from_numpy(hm).cuda(), None)
    else:
        press = np.asarray(press, np.float32)
        S = oracle.shifted_height_map(hm, press)
        Z, _ = taxim.deform(torch.from_numpy(hm).cuda(), torch.from_numpy(press).cuda())
    Zo, _ = oracle.gel_pad_deformation(S)
    ref, mag, _, im, idd = oracle.shade(Zo, True)
    assert np.abs(Z.cpu().numpy() - Zo).max() <= z_tol
    _, idx = taxim.shade(Z, return_bins=True)
    idx = idx.cpu().numpy().astype(np.int64)
    same = (idx[..., 0] == im) & (idx[..., 1] == idd)
    strong = mag > 1e-3
    stats = {"bin_equal_frac_all": float(same.mean())}
    if strong.any():
        stats["bin_equal_frac_strong"] = float(same[strong].mean())
        assert stats["bin_equal_frac_strong"] >= min_bin_frac, stats
    stats["rgb_rel_same_bin_max"] = float(rgb_rel_err(out_nhwc, ref)[same].max())
    assert stats["rgb_rel_same_bin_max"] <= rgb_tol, stats
    assert stats["bin_equal_frac_all"] >= 0.98, stats  # (flat pixels are deterministic on both sides: they agree too)
    return stats
