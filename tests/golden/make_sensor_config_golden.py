#!/usr/bin/env python3
"""Golden vectors of the reference's two SHIPPED sensor configurations in which camera, Taxim and FOTS resolutions differ
(container only; reads /root/reference at run time through _ref_harness.py, stores inputs + outputs):

  sensor_cfg_bench.npz       the benchmark harness (scripts/benchmarking/tactile_sim_performance/envs/ball_rolling_physx_rigid.py:161-199):
                             camera 320x240, clipping range (0.024, 0.034) m, Taxim 640x480 and FOTS 640x480 with a 9 x 11 marker
                             grid - the camera-resolution height map is up-sampled by BOTH simulators (TS:88-89, FS:121-122)
  sensor_cfg_taxim_fots.npz  the ball-rolling task (source/tacex_tasks/tacex_tasks/ball_rolling_tactile/ball_rolling_taxim_fots.py:300-331):
                             camera 32x24, clipping range (0.015, 0.029) m, Taxim 32x24, FOTS 320x240 on the 10x up-sampled height map

GelSightSensor itself imports isaaclab and cannot be loaded here, so its glue is restated statement by statement from
gelsight_sensor.py:342-378 (update order), 581-593 (depth -> height map: inf -> far clip, metres -> mm), taxim_sim.py:80-131
(resize, render, indentation depth) and fots_marker_sim.py:114-184 (resize, deformation with FOTS' own TaximTorch, marker loop);
the kernels underneath (TaximTorch, MarkerMotion, torchvision-style resize) are the reference's own code.

    python tests/golden/make_sensor_config_golden.py
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
from make_golden import CALIB_DST, GELPAD_HEIGHT, GELPAD_TO_CAMERA_MIN_DISTANCE  # noqa: E402
from tacex_amd.utils.synthetic import synthetic_depth_maps  # noqa: E402


def sensor_sequence(t, MarkerMotion, cam_res, clip, taxim_res, fots_res, grid, n, steps, seed, rgb_stride=1):
    """cam_res / taxim_res / fots_res are (width, height) like the reference cfgs; grid = (num_markers_col, num_markers_row)."""
    import torchvision.transforms.functional as F  # the harness shim: bilinear + antialias like torchvision >= 0.17

    Wc, Hc = cam_res
    Wt, Ht = taxim_res
    Wf, Hf = fots_res
    mm = MarkerMotion(frame0_blur=np.zeros((Hf, Wf, 3)), lamb=[0.00125, 0.00021, 0.00038], mm2pix=19.58, num_markers_col=grid[0],
                      num_markers_row=grid[1], tactile_img_width=Wf, tactile_img_height=Hf, x0=15, y0=26)
    init = np.stack((mm.init_marker_x_pos, mm.init_marker_y_pos), -1).reshape(-1, 2)
    hm0, _ = synthetic_depth_maps(n, Hc, Wc, seed=seed, flat_fraction=0.0)
    trajs = [[] for _ in range(n)]
    depths, hms, indents, rgbs, mds, ncs, ims, ids, strongs = [], [], [], [], [], [], [], [], []
    for s in range(steps):
        hm_mm = torch.roll(hm0, shifts=(s, 2 * s), dims=(1, 2)).clone()
        if s == 1 and n > 1:
            hm_mm[1] = 29.0  # env 1 lifts off in the second step (trajectory reset path)
        # what the TiledCamera hands over: metres, inf where nothing lies within the far clipping plane (GS:581-588)
        depth = hm_mm / 1000.0
        depth = torch.where(hm_mm >= 29.0, torch.full_like(depth, float("inf")), depth)
        depths.append(depth.numpy().copy())
        height_map = depth.clone()
        height_map[torch.isinf(height_map)] = clip[1]  # GS:586-588
        height_map *= 1000  # GS:590
        # TS:115-131 on the camera-resolution height map
        hmm = height_map / 1000
        d = hmm.amin((1, 2)) - GELPAD_TO_CAMERA_MIN_DISTANCE
        d = torch.where(d < 0, 0, d)
        indent = torch.where(d <= GELPAD_HEIGHT, (GELPAD_HEIGHT - d) * 1000, 0).float()
        # TS:80-113
        hm_t = height_map
        if (hm_t.shape[1], hm_t.shape[2]) != (Ht, Wt):
            hm_t = F.resize(hm_t, (Ht, Wt))
        rgb = t.render_direct(hm_t[:], with_shadow=False, press_depth=indent, orig_hm_fmt=False).movedim(1, 3)
        # the bins behind that image (private stages replayed like make_golden.py does): the parity protocol compares RGB on
        # same-bin pixels only - where the gel is flat the reference's direction bin is FFT round-off
        Zt, _ = t._TaximTorch__compute_gel_pad_deformation(t._TaximTorch__get_shifted_height_map(indent, hm_t))
        gm, gd = t._TaximTorch__generate_normals(-(Zt / t.sensor_params.pixmm))
        nb = t.sensor_params.num_bins
        ims.append(torch.floor(gm / (0.5 * torch.pi / (nb - 1))).numpy().astype(np.uint8))
        ids.append(torch.floor((gd + torch.pi) / (2 * torch.pi / (nb - 1))).numpy().astype(np.uint8))
        strongs.append(np.packbits((gm > 1e-3).numpy()))
        # FS:114-184
        hm_f = height_map
        if (hm_f.shape[1], hm_f.shape[2]) != (Hf, Wf):
            hm_f = F.resize(hm_f, (Hf, Wf))
        S = t._TaximTorch__get_shifted_height_map(indent, hm_f)
        Z, M = t._TaximTorch__compute_gel_pad_deformation(S)
        D = Z.max() - Z
        md = np.zeros((n, 2, init.shape[0], 2), np.float32)
        md[:, 0] = init
        nc = []
        for e in range(n):
            if indent[e].item() > 0.0:
                pts = torch.argwhere(M[e])
                mean = torch.mean(pts.float(), dim=0).cpu().numpy()
                mean[0] = (mean[0] - mm.tactile_img_height / 2) / mm.mm2pix
                mean[1] = (mean[1] - mm.tactile_img_width / 2) / mm.mm2pix
                trajs[e].append([mean[1], mean[0], 0.0])
                c = 0
                for i in range(mm.num_markers_col):
                    for j in range(mm.num_markers_row):
                        c += int(M[e].numpy()[int(mm.init_marker_y_pos[j, i]), int(mm.init_marker_x_pos[j, i])] == 1.0)
                nc.append(c)
                x, y = mm.marker_sim(D[e].cpu().numpy(), M[e].cpu().numpy(), trajs[e])
            else:
                trajs[e] = []
                nc.append(0)
                x, y = mm.init_marker_x_pos, mm.init_marker_y_pos
            md[e, 1] = np.stack((x, y), -1).reshape(-1, 2)
        hms.append(height_map.numpy().copy())
        indents.append(indent.numpy())
        rgbs.append(rgb[:, ::rgb_stride, ::rgb_stride].contiguous().numpy())  # every rgb_stride-th row / column (fixture size)
        mds.append(md)
        ncs.append(nc)
    return {
        "cam_res": np.array(cam_res), "clip": np.array(clip), "taxim_res": np.array(taxim_res), "fots_res": np.array(fots_res),
        "grid": np.array(grid), "rgb_stride": np.array(rgb_stride),
        "depth_m": np.stack(depths, 0), "height_map": np.stack(hms, 0), "indent": np.stack(indents, 0), "rgb": np.stack(rgbs, 0),
        "idx_mag": np.stack(ims, 0), "idx_dir": np.stack(ids, 0), "strong": np.stack(strongs, 0),
        "marker_data": np.stack(mds, 0), "n_contacts": np.array(ncs, np.int64), "init_marker_pos": init.astype(np.int64),
    }


def main():
    assert ref.reference_available(), "/root/reference is required to (re)generate the golden vectors"
    torch.manual_seed(0)
    torch.set_num_threads(8)
    t, _ = ref.load_reference_taxim(CALIB_DST)
    MarkerMotion = ref.load_reference_marker_motion()
    np.savez_compressed(HERE / "sensor_cfg_bench.npz",
                        **sensor_sequence(t, MarkerMotion, (320, 240), (0.024, 0.034), (640, 480), (640, 480), (9, 11), n=2, steps=2, seed=31, rgb_stride=2))
    np.savez_compressed(HERE / "sensor_cfg_taxim_fots.npz",
                        **sensor_sequence(t, MarkerMotion, (32, 24), (0.015, 0.029), (32, 24), (320, 240), (11, 9), n=3, steps=3, seed=32))
    for f in sorted(HERE.glob("sensor_cfg_*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
