"""Mesh depth source (arbitrary rigid indenters): the oracle against a closed form on CPU, the HIP rasteriser
against the oracle bit for bit on the GPU, and the source driving a GelSightSensor through cfg.sensor_camera_cfg.depth_source."""
import numpy as np
import pytest
import torch

from oracle.mesh_depth_oracle import icosphere, pose_rows, render_depth

INTR = dict(fx=340.0, fy=325.0, cx=160.0, cy=125.0)


def _analytic_sphere_depth(center, r, H, W, near, far, fx, fy, cx, cy):
    """z of the first ray / sphere intersection for every pixel centre (distance to the image plane), inf when missed."""
    j, i = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    d = np.stack([(j - cx) / fx, (i - cy) / fy, np.ones_like(j)], -1)  # ray direction with unit z: point = s * d, depth = s
    c = np.asarray(center, dtype=np.float64)
    a = (d * d).sum(-1); b = -2 * (d @ c); cc = c @ c - r * r
    disc = b * b - 4 * a * cc
    s = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    return np.where((s >= near) & (s <= far), s, np.inf)


def test_oracle_sphere_matches_closed_form():
    r, c = 0.004, (0.0012, -0.0008, 0.030)
    V, T = icosphere(r, 4)  # 2562 vertices / 5120 triangles
    pose = pose_rows(np.array([c]), np.array([[0.9, 0.1, -0.3, 0.2]]))  # the rotation must not matter for a sphere
    H, W = 120, 160
    intr = dict(fx=170.0, fy=162.5, cx=80.0, cy=62.5)
    got = render_depth(V, T, pose, H=H, W=W, near=0.02, far=0.04, **intr)[0]
    want = _analytic_sphere_depth(c, r, H, W, 0.02, 0.04, **intr)
    hit_g, hit_w = np.isfinite(got), np.isfinite(want)
    assert hit_w.sum() > 1500
    # the inscribed polyhedron is slightly smaller than the sphere: silhouettes differ by boundary pixels only
    assert (hit_g & ~hit_w).sum() == 0 and (hit_w & ~hit_g).sum() <= 0.03 * hit_w.sum()
    both = hit_g & hit_w
    # a level-4 icosphere's faces sit up to r (1 - cos 2.3 deg) = 3.2e-6 m inside the sphere; along the viewing ray that is
    # divided by the cosine of the incidence angle (<= 60 deg in the `inner` region, grazing near the silhouette)
    inner = both & (want < c[2] - 0.5 * r)
    err = np.abs(got[both] - want[both])
    assert np.abs(got[inner] - want[inner]).max() <= 1.5e-5 and np.abs(got[inner] - want[inner]).mean() <= 6e-6
    assert np.quantile(err, 0.99) <= 6e-5
    assert (got[both] >= want[both] - 1e-7).all()  # inscribed: never in front of the true surface


def test_oracle_clipping_and_culling():
    V = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float32) * 0.01
    T = np.array([[0, 1, 2], [0, 3, 2]], dtype=np.int32)  # opposite windings: both must render
    kw = dict(H=48, W=64, fx=68.0, fy=65.0, cx=32.0, cy=25.0)
    d = render_depth(V, T, pose_rows(np.array([[0, 0, 0.026]]), np.array([[1.0, 0, 0, 0]])), near=0.024, far=0.029, **kw)[0]
    assert np.isfinite(d).sum() > 500 and np.allclose(d[np.isfinite(d)], 0.026, atol=1e-7)
    # tilted about x: depth varies linearly in the row direction, fragments outside [near, far] are clipped per pixel
    q = np.array([[np.cos(0.35), np.sin(0.35), 0, 0]])
    d = render_depth(V, T, pose_rows(np.array([[0, 0, 0.0265]]), q), near=0.024, far=0.029, **kw)[0]
    f = np.isfinite(d)
    assert 0 < f.sum() < kw["H"] * kw["W"] and d[f].min() >= 0.024 and d[f].max() <= 0.029 and d[f].max() - d[f].min() > 0.003
    # behind the camera / beyond the far plane: nothing
    assert not np.isfinite(render_depth(V, T, pose_rows(np.array([[0, 0, -0.03]]), np.array([[1.0, 0, 0, 0]])), near=0.0, far=1.0, **kw)).any()
    assert not np.isfinite(render_depth(V, T, pose_rows(np.array([[0, 0, 0.05]]), np.array([[1.0, 0, 0, 0]])), near=0.024, far=0.029, **kw)).any()


@pytest.mark.gpu
def test_hip_rasteriser_equals_oracle_and_drives_the_sensor(calib_dir):
    from tacex_amd import GelSightSensor, GelSightSensorCfg, MeshDepthSource
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg

    B, H, W = 6, 240, 320
    Vs, Ts = icosphere(0.004, 3)
    # a second, offset lobe makes the mesh non-convex (overlapping surfaces: the z-buffer must keep the nearer one)
    V = np.concatenate([Vs, Vs * 0.6 + np.array([0.003, 0.001, -0.0015], dtype=np.float32)])
    T = np.concatenate([Ts, Ts + len(Vs)])
    src = MeshDepthSource(V, T, B, "cuda:0", resolution=(W, H), intrinsics=tuple(INTR.values()), clipping_range=(0.024, 0.029))
    rng = np.random.RandomState(4)
    pos = np.stack([rng.uniform(-0.004, 0.004, B), rng.uniform(-0.003, 0.003, B), rng.uniform(0.0290, 0.0315, B)], 1)
    pos[4] = [0.013, 0.0, 0.030]   # half out of view on the right
    pos[5] = [0.0, 0.0, 0.040]     # beyond the far plane: no contact at all
    quat = rng.normal(size=(B, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    src.pos.copy_(torch.from_numpy(pos).float()); src.quat.copy_(torch.from_numpy(quat).float())
    depth = src().cpu().numpy()
    want = render_depth(V, T, pose_rows(src.pos.cpu().numpy(), src.quat.cpu().numpy()), H=H, W=W, near=0.024, far=0.029, **INTR)
    np.testing.assert_array_equal(np.isfinite(depth), np.isfinite(want))
    m = np.isfinite(want)
    np.testing.assert_array_equal(depth[m], want[m])  # same float32 operations in the same order
    assert m[:4].reshape(4, -1).sum(1).min() > 300 and m[4].sum() > 50 and m[5].sum() == 0
    # ... and through the sensor: depth_source replaces the TiledCamera (GS:357-359), the rest of the path is unchanged
    cfg = GelSightSensorCfg(
        num_envs=B, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029), depth_source=src),
        data_types=["tactile_rgb", "height_map"],
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(calib_dir), gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,
                                          tactile_img_res=(W, H), device="cuda:0"),
        marker_motion_sim_cfg=None, device="cuda:0")
    s = GelSightSensor(cfg); s.initialize()
    s.update(0.01, force_recompute=True)
    hm = s.data.output["height_map"].cpu().numpy()
    np.testing.assert_array_equal(hm, np.where(np.isfinite(want), want, np.float32(0.029)) * np.float32(1000.0))  # GS:585-590
    ind = s.indentation_depth.cpu().numpy()
    assert (ind[:4] > 0.1).all() and ind[5] == 0.0
    rgb = s.data.output["tactile_rgb"]
    bg = s.optical_simulator.background_img if hasattr(s.optical_simulator, "background_img") else None
    assert torch.isfinite(rgb).all() and (rgb[0] - rgb[5]).abs().max() > 0.02  # a pressed frame differs from the untouched one
