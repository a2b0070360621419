#!/usr/bin/env python3
"""Where do the shadow-branch pixels with |rgb - reference| > 1e-3 come from?  (container only: imports the reference.)

The shadow render is  clip(blur5(blur3(min(shade, shadow_map)) + background))  (TT:330-346 at 320x240).  This study runs the
reference on the 240x320 fixture with its scatter_min call intercepted (the shadow map BEFORE the two blurs), runs the oracle on
the same input, and attributes every pixel of the final image that differs by more than 1e-3 - inside the region the GPU test
compares (tests/parity.py:well_conditioned_field) - to a PRE-blur difference inside its 7x7 receptive field:

  * a shadow-map sample that landed on a different pixel / is missing (float32 cos/sin times step, truncated by .long(), sits on an
    integer boundary; or the ring pixel's direction bin differs), or
  * a shade pixel whose (magnitude, direction) bin differs.

With gpurun_out/shadow_dump.npz (scripts/shadow_dump.py run on the GPU box: the HIP path's deformed gel, mask, bins and RGB of the
same fixture) the same attribution is made for the HIP path, and the oracle is fed the HIP path's deformed gel to show that the
moved samples follow from the last-bit differences of Z alone.

    python tests/studies/shadow_outliers.py      # prints the table kept in DESIGN.md section 2
"""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "tests" / "golden"))

import _ref_harness as ref  # noqa: E402
from make_golden import CALIB_DST  # noqa: E402
from oracle.taxim_oracle import TaximOracle  # noqa: E402
from parity import well_conditioned_field  # noqa: E402
from scipy import ndimage  # noqa: E402


def attribute(name, rgb, im, idd, sh, rgb_ref, sh_ref, g, dump_example=True):
    B = rgb.shape[0]
    ok = well_conditioned_field(im, idd, g)
    fin = np.isfinite(sh_ref) | np.isfinite(sh)
    both = np.isfinite(sh_ref) & np.isfinite(sh)
    sample_diff_px = ((np.isfinite(sh_ref) != np.isfinite(sh)) | (both & (sh_ref != sh))).any(-1)
    bin_diff_px = (im != g["idx_mag"]) | (idd != g["idx_dir"])
    d = np.abs(rgb - rgb_ref).max(-1)
    out = (d > 1e-3) & ok
    st = np.ones((7, 7), bool)  # k = 3 then k = 5: 7x7 receptive field
    near_sample = np.stack([ndimage.binary_dilation(sample_diff_px[b], structure=st) for b in range(B)])
    near_bin = np.stack([ndimage.binary_dilation(bin_diff_px[b], structure=st) for b in range(B)])
    print(f"--- {name} vs the reference")
    print(f"compared (well-conditioned) pixels {int(ok.sum())}; shadow-map pixels with a sample {int(fin.any(-1).sum())}")
    print(f"shadow-map pixels whose sample differs or is missing on one side: {int(sample_diff_px.sum())} "
          f"({sample_diff_px.sum() / max(1, fin.any(-1).sum()):.2%} of the sampled pixels)")
    print(f"final-image pixels > 1e-3 inside the compared region: {int(out.sum())} ({out.sum() / ok.sum():.3%})")
    print(f"  with a differing shadow sample in their 7x7 field : {int((out & near_sample).sum())}")
    print(f"  with a differing shade bin in their 7x7 field     : {int((out & near_bin).sum())}")
    print(f"  with neither                                      : {int((out & ~near_sample & ~near_bin).sum())}")
    rest = ok & ~near_sample & ~near_bin
    print(f"max |d| over compared pixels with NO pre-blur difference in their field: {d[rest].max():.2e} ({int(rest.sum())} pixels)")
    if dump_example and out.any():
        b, y, x = np.unravel_index(np.argmax(np.where(out, d, 0)), d.shape)
        print(f"worst outlier: frame {b} pixel (y={y}, x={x}), |d| = {d[b, y, x]:.4f}; reference rgb {rgb_ref[b, y, x]}, {name} {rgb[b, y, x]}")
        y0, x0 = max(0, y - 3), max(0, x - 3)
        ys, xs = np.nonzero(sample_diff_px[b, y0:y + 4, x0:x + 4])
        for yy, xx in zip(ys + y0, xs + x0):
            print(f"  shadow map at (y={yy}, x={xx}): reference {sh_ref[b, yy, xx]}, {name} {sh[b, yy, xx]}")
    return sample_diff_px


def main():
    g = dict(np.load(REPO / "tests/golden/taxim_240x320.npz"))
    t, _ = ref.load_reference_taxim(CALIB_DST)
    ts = sys.modules["torch_scatter"]
    captured = {}
    inner = ts.scatter_min

    def spy(src, index, dim_size=None, out=None):
        r = inner(src, index, dim_size=dim_size, out=out)
        captured["shadow"] = out.clone()
        return r

    ts.scatter_min = spy
    hm, indent = torch.from_numpy(g["hm"]), torch.from_numpy(g["indent"])
    rgb_ref = t.render_direct(hm, with_shadow=True, press_depth=indent, orig_hm_fmt=False).movedim(1, 3).numpy()
    ts.scatter_min = inner
    B, H, W = g["hm"].shape
    sh_ref = captured["shadow"].reshape(3, B, H, W).movedim(0, -1).numpy()
    assert np.abs(rgb_ref - g["rgb_shadow"]).max() == 0.0  # the fixture is this very render

    o = TaximOracle(CALIB_DST, (H, W), "direct")
    Z, M = o.gel_pad_deformation(o.shifted_height_map(g["hm"], g["indent"]))
    sh_or, _ = o.shadow_map(Z, M)
    rgb_or = o.shade_with_shadow(Z, M)
    mag, dr = o.normals(-(Z / np.float32(o.p.pixmm)))
    im, idd = o.bins(mag, dr)
    attribute("oracle", rgb_or, im, idd, sh_or, rgb_ref, sh_ref, g)

    dump = REPO / "gpurun_out/shadow_dump.npz"
    if not dump.exists():
        print("(no gpurun_out/shadow_dump.npz: run scripts/shadow_dump.py on the GPU box for the HIP path's attribution)")
        return
    dmp = np.load(dump)
    Zg = dmp["Z"]
    Mg = np.unpackbits(dmp["M"])[:Zg.size].reshape(Zg.shape).astype(bool)
    img, idg = dmp["idx"][..., 0].astype(np.int64), dmp["idx"][..., 1].astype(np.int64)
    print(f"\nHIP path: max |Z - Z_reference| = {np.abs(Zg.astype(np.float64) - g['Z']).max():.2e} mm, contact mask differs on {int((Mg != M).sum())} pixels")
    # the oracle's ray march on the HIP path's deformed gel = the HIP path's shadow map (test_shadow_ray_samples_exact_vs_oracle)
    sh_g, gdir_g = o.shadow_map(Zg, Mg)
    _, gdir_o = o.shadow_map(Z, M)
    moved = attribute("HIP", dmp["rgb"], img, idg, sh_g, rgb_ref, sh_ref, g)
    rgb_or_g = o.shade_with_shadow(Zg, Mg)
    print(f"oracle fed the HIP path's Z / mask vs the HIP render: max |d| = {np.abs(rgb_or_g - dmp['rgb']).max():.2e}")
    dil = Mg.astype(np.float32)
    for (kw, kh) in o.shadow_attachment_rounds():
        dil = o._box_dilate_same(dil, int(kh), int(kw))
    ring = (dil != 0) & ~Mg & ~M
    prec = np.float32(o.p.sim["discretize_precision"])
    nb = np.floor((gdir_g + np.float32(np.pi)) / prec) != np.floor((gdir_o + np.float32(np.pi)) / prec)
    mag_o, _ = o.normals(-(Z / np.float32(o.p.pixmm)))
    sel = nb & ring
    print(f"ring pixels (ray sources): {int(ring.sum())}; their shadow DIRECTION bin differs between the two Z on {int(sel.sum())}"
          + (f"; gradient magnitude at those: median {np.median(mag_o[sel]):.1e}, max {mag_o[sel].max():.1e}" if sel.any() else "")
          + f" (ring gradient magnitude: median {np.median(mag_o[ring]):.1e}, min {mag_o[ring].min():.1e})")
    print(f"shadow-map pixels that moved: {int(moved.sum())}")


if __name__ == "__main__":
    main()
