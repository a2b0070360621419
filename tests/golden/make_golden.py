#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (container only).

    python tests/golden/make_golden.py

Reads /root/reference at run time (never copies its source): imports the reference's TaximTorch and
MarkerMotion through _ref_harness.py, feeds them seeded synthetic depth maps and stores inputs +
outputs (+ intermediates of the private stages, reached name-mangled exactly like the reference's own
FOTS wrapper does, fots_marker_sim.py:128-129) as compressed .npz files.

Also (re)creates the calibration folder shipped with the package
(tacex_amd/assets/calib/gsmini_640x480): the reference's real calibration *data* files plus a
synthesized dataPack.npz (the real background frame is absent from the mount).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

import _ref_harness as ref  # noqa: E402
from tacex_amd.utils.synthetic import synthetic_depth_maps  # noqa: E402

CALIB_DST = REPO / "tacex_amd" / "assets" / "calib" / "gsmini_640x480"
GELPAD_HEIGHT = 0.0045
GELPAD_TO_CAMERA_MIN_DISTANCE = 0.024


def indentation_depth(hm_mm: torch.Tensor) -> torch.Tensor:
    """taxim_sim.py:115-131 restated inline (the wrapper itself imports omni and cannot be loaded)."""
    hm = hm_mm / 1000
    d = hm.amin((1, 2)) - GELPAD_TO_CAMERA_MIN_DISTANCE
    d = torch.where(d < 0, 0, d)
    return torch.where(d <= GELPAD_HEIGHT, (GELPAD_HEIGHT - d) * 1000, 0).float()


def taxim_case(t, H, W, n, seed, levels: bool, shadow: bool, kinds=None, slim: bool = False, n_levels_frames=None,
               shadow_frames=None):
    kw = {} if kinds is None else {"kinds": kinds}
    hm, _ = synthetic_depth_maps(n, H, W, seed=seed, flat_fraction=0.0, **kw)
    hm[-1] = 29.0  # last frame: no contact at all
    if n > 2:
        hm[-2] = 28.6  # flat object hovering above the gel (min > gel top -> indent 0)
    indent = indentation_depth(hm)
    S = t._TaximTorch__get_shifted_height_map(indent, hm)
    gel = t._TaximTorch__get_gel_map_cached((H, W))
    out = {
        "hm": hm.numpy(),
        "indent": indent.numpy(),
        "S": S.numpy(),
        "gel": gel.numpy(),
        "bg": t._TaximTorch__get_background_img_cached((H, W)).numpy(),
    }
    Z, M = t._TaximTorch__compute_gel_pad_deformation(S)
    out["Z"] = Z.numpy()
    out["M"] = np.packbits(M.numpy())
    if levels:
        # replay the pyramid stage by stage with the reference's own blur (TT:464-471)
        J = torch.minimum(S, gel)
        Zl = J
        lv = []
        for sg in zip(*t.sim_params.deform_pyramid_sigma((H, W))):
            Zl = t._TaximTorch__gaussian_blur(Zl.unsqueeze(0), sg)[0]
            Zl[M] = J[M]
            lv.append(Zl.clone().numpy())
        out["J"] = J.numpy()
        zl = np.stack(lv, 0)  # (6, B, H, W)
        out["Z_levels"] = zl if n_levels_frames is None else zl[:, :n_levels_frames]
    gm, gd = t._TaximTorch__generate_normals(-(Z / t.sensor_params.pixmm))
    nb = t.sensor_params.num_bins
    im = torch.floor(gm / (0.5 * torch.pi / (nb - 1))).long()
    idd = torch.floor((gd + torch.pi) / (2 * torch.pi / (nb - 1))).long()
    out["grad_mag"] = gm.numpy()
    out["grad_dir"] = gd.numpy()
    out["idx_mag"] = im.numpy().astype(np.uint8)
    out["idx_dir"] = idd.numpy().astype(np.uint8)
    rgb = t.render_direct(hm, with_shadow=False, press_depth=indent, orig_hm_fmt=False)
    out["rgb"] = rgb.movedim(1, 3).contiguous().numpy()
    if slim:  # big resolutions: keep only what the parity protocol needs
        for k in ("S", "bg", "grad_mag", "grad_dir"):
            out.pop(k)
    if shadow:
        rgbs = t.render_direct(hm, with_shadow=True, press_depth=indent, orig_hm_fmt=False)
        rgbs = rgbs.movedim(1, 3).contiguous().numpy()
        out["rgb_shadow"] = rgbs if shadow_frames is None else rgbs[:shadow_frames]  # frames render independently
    return out


def kernel_tables(t, mod):
    """Gaussian sizes + float32 taps the reference derives for each resolution (TT:362-403)."""
    T = mod.TaximTorch
    res = {}
    for (H, W) in [(240, 320), (480, 640), (32, 32), (24, 32), (48, 64)]:
        pw, ph = t.sim_params.deform_pyramid_sigma((H, W))
        fw, fh = t.sim_params.deform_final_sigma((H, W))
        sig = list(zip(pw, ph)) + [(fw, fh)]
        eps = 1e-5
        ks = []
        for s in sig:
            s_np = np.array(s)
            k = (np.round(np.sqrt(-2 * np.log(eps * np.sqrt(2 * np.pi) * s_np)) * s_np).astype(np.int_) // 2 * 2 + 1)
            ks.append(k.tolist())
        res[f"sigma_{H}x{W}"] = np.array(sig, dtype=np.float64)
        res[f"ksize_{H}x{W}"] = np.array(ks, dtype=np.int64)
        for li, (s, k) in enumerate(zip(sig, ks)):
            res[f"taps_w_{H}x{W}_{li}"] = T._TaximTorch__get_gaussian_kernel1d(s[0], k[0]).numpy()
            res[f"taps_h_{H}x{W}_{li}"] = T._TaximTorch__get_gaussian_kernel1d(s[1], k[1]).numpy()
    return res


def fots_case(t, MarkerMotion, seed: int, n: int, steps: int, H: int = 240, W: int = 320, ncol: int = 11, nrow: int = 9):
    hm0, _ = synthetic_depth_maps(n, H, W, seed=seed, flat_fraction=0.0)
    mm = MarkerMotion(
        frame0_blur=np.zeros((H, W, 3)),
        lamb=[0.00125, 0.00021, 0.00038],
        mm2pix=19.58,
        num_markers_col=ncol,
        num_markers_row=nrow,
        tactile_img_width=W,
        tactile_img_height=H,
        x0=15,
        y0=26,
    )
    init = np.stack((mm.init_marker_x_pos, mm.init_marker_y_pos), -1).reshape(-1, 2)
    trajs = [[] for _ in range(n)]
    hms, thetas, outs, indents, ncontacts, csets = [], [], [], [], [], []
    for s in range(steps):
        # the indenter drifts sideways and (in the middle step, env 1) lifts off -> traj reset path
        hm = torch.roll(hm0, shifts=(2 * s, 3 * s), dims=(1, 2)).clone()
        if s == 2:
            hm[1] = 29.0
        theta = np.array([0.02 * s * (1 + e) for e in range(n)], dtype=np.float32)
        indent = indentation_depth(hm)
        S = t._TaximTorch__get_shifted_height_map(indent, hm)
        Z, M = t._TaximTorch__compute_gel_pad_deformation(S)
        D = Z.max() - Z  # fots_marker_sim.py:130
        md = np.zeros((n, 2, init.shape[0], 2), np.float32)
        md[:, 0] = init
        nc = []
        cset = np.zeros((n, init.shape[0]), np.uint8)  # contact SET: markers (row-major j * ncol + i) whose pixel is in the mask
        for e in range(n):
            if indent[e].item() > 0.0:  # fots_marker_sim.py:133-175
                pts = torch.argwhere(M[e])
                mean = torch.mean(pts.float(), dim=0).cpu().numpy()
                mean[0] = (mean[0] - mm.tactile_img_height / 2) / mm.mm2pix
                mean[1] = (mean[1] - mm.tactile_img_width / 2) / mm.mm2pix
                trajs[e].append([mean[1], mean[0], theta[e]])
                # count contacts the way MM:152-166 does, for the fixture
                c = 0
                for i in range(mm.num_markers_col):
                    for j in range(mm.num_markers_row):
                        if M[e].numpy()[int(mm.init_marker_y_pos[j, i]), int(mm.init_marker_x_pos[j, i])] == 1.0:
                            c += 1
                            cset[e, j * mm.num_markers_col + i] = 1
                nc.append(c)
                x, y = mm.marker_sim(D[e].cpu().numpy(), M[e].cpu().numpy(), trajs[e])
            else:
                trajs[e] = []
                nc.append(0)
                x, y = mm.init_marker_x_pos, mm.init_marker_y_pos
            md[e, 1] = np.stack((x, y), -1).reshape(-1, 2)
        hms.append(hm.numpy())
        thetas.append(theta)
        outs.append(md)
        indents.append(indent.numpy())
        ncontacts.append(nc)
        csets.append(cset)
    return {
        "hm": np.stack(hms, 0),
        "theta": np.stack(thetas, 0),
        "indent": np.stack(indents, 0),
        "marker_data": np.stack(outs, 0),
        "n_contacts": np.array(ncontacts, np.int64),
        "contact_set": np.stack(csets, 0),
        "init_marker_pos": init.astype(np.int64),
        "marker_x_idx": mm.marker_x_idx.astype(np.int64),
        "marker_y_idx": mm.marker_y_idx.astype(np.int64),
    }


def main():
    assert ref.reference_available(), "/root/reference is required to (re)generate the golden vectors"
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref.make_calib_dir(CALIB_DST, seed=7)
    (CALIB_DST / "README.md").write_text(
        "Calibration DATA for the GelSight Mini (640x480 calibration resolution).\n\n"
        "params.json, polycalib.npz, gelmap.npy, shadowTable.npz: the reference's calibration tables\n"
        "(tacex_assets/data/Sensors/GelSight_Mini/calibs/640x480), byte-identical data files.\n"
        "dataPack.npz: SYNTHESIZED background frame f0 (seed 7, tests/golden/_ref_harness.py:synth_f0) -\n"
        "the real one is absent from the reference checkout (.MISSING_LARGE_BLOBS).\n"
    )
    t, mod = ref.load_reference_taxim(CALIB_DST)
    MarkerMotion = ref.load_reference_marker_motion()

    np.savez_compressed(HERE / "taxim_tables.npz", **kernel_tables(t, mod))
    np.savez_compressed(HERE / "taxim_32x32.npz", **taxim_case(t, 32, 32, 4, 11, levels=True, shadow=True))
    np.savez_compressed(HERE / "taxim_24x32.npz", **taxim_case(t, 24, 32, 3, 12, levels=True, shadow=False))
    np.savez_compressed(HERE / "taxim_48x64.npz", **taxim_case(t, 48, 64, 4, 13, levels=True, shadow=True))
    np.savez_compressed(HERE / "taxim_240x320.npz", **taxim_case(t, 240, 320, 5, 14, levels=True, shadow=True, n_levels_frames=2))
    np.savez_compressed(HERE / "taxim_480x640.npz", **taxim_case(t, 480, 640, 2, 15, levels=False, shadow=True, slim=True, shadow_frames=1))
    np.savez_compressed(HERE / "fots_240x320.npz", **fots_case(t, MarkerMotion, seed=21, n=4, steps=4))
    # the marker grid of the reference's 640x480 benchmark (ball_rolling_physx_rigid.py:172-179: 9 columns x 11 rows)
    np.savez_compressed(HERE / "fots_480x640.npz", **fots_case(t, MarkerMotion, seed=22, n=2, steps=3, H=480, W=640, ncol=9, nrow=11))
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
