"""GPU tests of the gelpad FEM kernels against the CPU oracle (oracle/fem_oracle.py; parity with libuipc UNPINNED)."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sim(points, tets, B, strength=100.0, dt=0.01):
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg

    sim = UipcSim(UipcSimCfg(device="cuda:0", dt=dt), num_envs=B)
    sim.cfg.linear_system.coarse_grid = None  # these tests pin the block-Jacobi path; the C4 tests below run the two-level one
    sim.cfg.linear_system.vertex_chains = None
    UipcObject(UipcObjectCfg(mesh_points=points, mesh_tets=tets), sim)
    sim.setup_sim(constraint_strength_ratio=strength)
    return sim


@pytest.fixture(scope="module")
def meshes(golden_dir):
    return np.load(golden_dir / "fem_meshes.npz")


@pytest.mark.parametrize("name", ["cube", "simple_axle", "link"])
def test_element_terms_vs_oracle(meshes, name):
    from oracle.fem_oracle import FemModel

    P, Tt = meshes[f"{name}_points"], meshes[f"{name}_tets"]
    m = FemModel.build(P, Tt, youngs=1e4)
    B = 3
    rng = np.random.default_rng(0)
    x = np.stack([m.X, m.X + 0.03 * np.ptp(m.X) * rng.normal(size=m.X.shape),
                  m.X * np.array([0.6, 1.2, 0.8]) + 0.02 * np.ptp(m.X) * rng.normal(size=m.X.shape)])
    sim = _sim(P, Tt, B)
    xd = torch.from_numpy(x).cuda()
    e, g, h = sim.element_terms(xd)
    eo, go, ho = m.element_energy(x), m.element_gradient(x), m.element_hessian(x)
    sc_e, sc_g, sc_h = np.abs(eo).max() + 1e-300, np.abs(go).max(), np.abs(ho).max()
    assert np.abs(e.cpu().numpy() - eo).max() <= 1e-11 * sc_e + 1e-22
    assert np.abs(g.cpu().numpy().transpose(0, 2, 1) - go).max() <= 1e-11 * sc_g
    hh = h.cpu().numpy().reshape(B, 12, 12, -1).transpose(0, 3, 1, 2)
    assert np.abs(hh - ho).max() <= 1e-11 * sc_h
    # rest state (env 0): zero force
    assert np.abs(g[0].cpu().numpy()).max() <= 1e-9 * sc_g
    # PSD projection: eigenvalues >= 0, PSD elements unchanged
    _, _, hp = sim.element_terms(xd, energy=False, gradient=False, project_psd=True)
    hp = hp.cpu().numpy().reshape(B, 12, 12, -1).transpose(0, 3, 1, 2)
    hpo = m.element_hessian(x, project_psd=True)
    assert np.abs(hp - hpo).max() <= 1e-8 * sc_h
    assert np.linalg.eigvalsh(0.5 * (hp + hp.transpose(0, 1, 3, 2))).min() >= -1e-9 * sc_h


def test_energy_gradient_vs_oracle(meshes):
    from oracle.fem_oracle import FemModel

    P, Tt = meshes["simple_axle_points"], meshes["simple_axle_tets"]
    m = FemModel.build(P, Tt, youngs=1e4, strength=250.0)
    rng = np.random.default_rng(1)
    B, L = 4, np.ptp(P)
    x = m.X[None] + 0.02 * L * rng.normal(size=(B,) + m.X.shape)
    xt = m.X[None] + 0.005 * L * rng.normal(size=(B,) + m.X.shape)
    cons = (rng.random((B, len(P))) < 0.2)
    aim = m.X[None] + 0.01 * L * rng.normal(size=(B,) + m.X.shape)
    sim = _sim(P, Tt, B, strength=250.0)
    sim.x = torch.from_numpy(x).cuda()
    sim.x_tilde = torch.from_numpy(xt).cuda()
    sim.is_constrained = torch.from_numpy(cons.astype(np.uint8)).cuda()
    sim.aim_position = torch.from_numpy(aim).cuda()
    for b_cons in (True, False):
        E = sim.energy(constrained=b_cons).cpu().numpy()
        g = sim.gradient(constrained=b_cons).cpu().numpy()
        for b in range(B):
            c = cons[b].astype(np.float64) if b_cons else None
            a = aim[b] if b_cons else None
            Eo = m.energy(x[b], xt[b], c, a)
            go = m.gradient(x[b], xt[b], c, a)
            assert abs(E[b] - Eo) <= 1e-11 * abs(Eo)
            assert np.abs(g[b] - go).max() <= 1e-10 * np.abs(go).max()


def test_newton_step_vs_oracle_and_monotone():
    from oracle.fem_oracle import FemModel, box_tet_mesh

    P, Tt = box_tet_mesh(4, 5, 2)
    m = FemModel.build(P, Tt, youngs=1e4, strength=100.0)
    B = 3
    top = np.where(P[:, 2] > P[:, 2].max() - 1e-9)[0]
    sim = _sim(P, Tt, B)
    sim.cfg.linear_system.max_iter = 200
    sim.cfg.linear_system.tol_rate = 1e-8
    depths = [0.0004, 0.0008, 0.0012]
    aim = torch.from_numpy(P[top]).cuda()[None].repeat(B, 1, 1)
    for b, d in enumerate(depths):
        aim[b, :, 2] -= d
    sim.set_constraints(top, aim)
    sim.x_tilde = sim.x + sim.cfg.dt**2 * torch.tensor([0, 0, -9.8], dtype=torch.float64, device="cuda")
    xo = [P.copy() for _ in range(B)]
    xt = P + m.dt**2 * np.array([0, 0, -9.8])
    cons = np.zeros(len(P)); cons[top] = 1.0
    E_prev = None
    for it in range(6):
        st = sim.newton_step().cpu().numpy().copy()
        assert (st[:, 1] <= st[:, 0] + 1e-18).all(), "line search must not increase the energy"
        if E_prev is not None:
            np.testing.assert_allclose(st[:, 0], E_prev, rtol=1e-10)
        E_prev = st[:, 1]
        for b in range(B):
            aim_b = P.copy(); aim_b[top, 2] -= depths[b]
            xo[b], so = m.newton_step(xo[b], xt, cons, aim_b, pcg_max_iter=200, pcg_tol_rate=1e-8)
            assert abs(st[b, 0] - so[0]) <= 1e-8 * abs(so[0]) + 1e-20, (it, b)
            assert abs(st[b, 1] - so[1]) <= 1e-6 * abs(so[1]) + 1e-20, (it, b)
            if so[0] - so[1] > 1e-7 * abs(so[0]):  # at convergence accept/reject is decided by roundoff
                assert st[b, 2] == so[2]
    x = sim.x.cpu().numpy()
    for b in range(B):
        assert np.abs(x[b] - xo[b]).max() <= 1e-6 * np.ptp(P)


def test_lds_and_streaming_newton_kernels_agree(tmp_path):
    """The CU-resident Newton kernel (default) and the streaming one (TACEX_FEM_NEWTON_LDS=0) solve the same system; they
    differ only in the summation order of the nodal gathers (tet renumbering).  The switch is read once per process."""
    import os
    import subprocess
    import sys

    from conftest import REPO

    script = tmp_path / "newton_run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg\n"
        "from tacex_amd.uipc.uipc_object import gelpad_box_mesh\n"
        "P, T = gelpad_box_mesh(8, 10, 4)\n"
        "B = 4\n"
        "sim = UipcSim(UipcSimCfg(device='cuda:0'), num_envs=B)\n"
        "sim.cfg.linear_system.coarse_grid = None  # the streaming kernel has no coarse correction: compare like with like\n"
        "sim.cfg.linear_system.vertex_chains = None\n"
        "UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)\n"
        "sim.setup_sim()\n"
        "sim.cfg.linear_system.max_iter = 60\n"
        "top = np.where(P[:, 2] > P[:, 2].max() - 1e-9)[0]\n"
        "aim = torch.from_numpy(P[top]).cuda()[None].repeat(B, 1, 1)\n"
        "aim[:, :, 2] -= torch.linspace(0.0002, 0.0012, B, device='cuda', dtype=torch.float64)[:, None]\n"
        "sim.set_constraints(top, aim); sim.x_tilde = sim.x.clone()\n"
        "st = [sim.newton_step().cpu().numpy().copy() for _ in range(3)]\n"
        "np.savez(sys.argv[1], x=sim.x.cpu().numpy(), st=np.stack(st))\n")
    outs = {}
    for flag in ("1", "0"):
        out = tmp_path / f"n{flag}.npz"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, TACEX_FEM_NEWTON_LDS=flag),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = np.load(out)
    np.testing.assert_array_equal(outs["1"]["st"][..., 3], outs["0"]["st"][..., 3])  # same PCG iteration counts
    np.testing.assert_allclose(outs["1"]["st"][..., :2], outs["0"]["st"][..., :2], rtol=1e-9)
    scale = np.ptp(outs["0"]["x"])
    assert np.abs(outs["1"]["x"] - outs["0"]["x"]).max() <= 1e-9 * scale


def test_step_api_and_marker_uv(meshes):
    from oracle.fem_oracle import marker_uv
    from tacex_amd import _lib
    from tacex_amd.uipc import UipcObject, UipcObjectCfg
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    P, Tt = gelpad_box_mesh(4, 5, 2)
    sim = _sim(P, Tt, 2)
    sim.cfg.newton.velocity_tol = 1e-3
    bottom = np.where(P[:, 2] < 1e-12)[0]
    sim.set_constraints(bottom, torch.from_numpy(P[bottom]).cuda()[None].repeat(2, 1, 1))
    x = sim.step(max_newton_iter=30)
    assert torch.isfinite(x).all()
    assert sim.last_newton_iters <= 30
    # gravity pulls the free top down a little, the glued bottom stays
    free_fall = 0.5 * 9.8 * sim.cfg.dt**2 * 2  # dt^2 g
    assert x[:, bottom, 2].abs().max() < 0.2 * free_fall  # soft constraint holds the glued face back
    assert x[:, :, 2].max() <= P[:, 2].max() + 1e-9
    obj = sim.uipc_objects[0]
    tri = obj.surface_triangles()[:50]
    rng = np.random.default_rng(0)
    w = rng.dirichlet(np.ones(3), size=len(tri))
    cam = x.clone(); cam[..., 2] += 0.02
    uv = torch.empty((2, len(tri), 2), dtype=torch.float64, device="cuda")
    lib = _lib.load_library()
    tri_d = torch.from_numpy(tri).cuda(); w_d = torch.from_numpy(w).cuda()
    _lib.check(lib.tacex_fem_marker_uv(cam.data_ptr(), tri_d.data_ptr(), w_d.data_ptr(), 340.0, 325.0, 160.0, 125.0,
                                       uv.data_ptr(), 2, cam.shape[1], len(tri), torch.cuda.current_stream().cuda_stream), "uv")
    np.testing.assert_allclose(uv.cpu().numpy(), marker_uv(cam.cpu().numpy(), tri, w), rtol=1e-12)


def test_mani_skill_marker_flow_plugin():
    """GelSightSensor + ManiSkillSimulator (FEM-driven markers, MS:22-86 / VT:354-413) for several envs at once:
    barycentric surface points + pinhole projection on the GPU vs the NumPy oracle; mask/pad semantics of VT:382-405."""
    from oracle.fem_oracle import marker_uv
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.simulation_approaches.fem_based import ManiSkillSimulatorCfg
    from tacex_amd.simulation_approaches.fem_based.sim.tactile_sensor_uipc import gen_marker_grid, gen_marker_weight
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    B = 3
    P, Tt = gelpad_box_mesh(10, 8, 3, size=(0.030, 0.018, 0.0045))
    P = P - np.array([0.011, 0.009, 0.0])  # marker area over the camera axis
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
    gel = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=Tt), sim)
    sim.setup_sim()
    cfg = GelSightSensorCfg(
        num_envs=B, data_types=["marker_motion"], optical_sim_cfg=None,
        marker_motion_sim_cfg=ManiSkillSimulatorCfg(device="cuda:0", camera_pos_w=(0.0, 0.0, -0.024)),
        sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(320, 240)), device="cuda:0")
    cfg.compute_indentation_depth_class = "marker_motion_sim"
    sensor = GelSightSensor(cfg, gelpad_obj=gel)
    sensor.compute_indentation_depth_func = None
    sensor.initialize()
    sensor.compute_indentation_depth_func = None
    ms = sensor.marker_motion_simulator.marker_motion_sim
    # deform every env differently (smooth displacement field), then read the flow through the sensor
    x = sim.x.clone()
    for b in range(B):
        x[b, :, 0] += 0.0004 * (b + 1) * torch.sin(300 * x[b, :, 1])
        x[b, :, 2] += 0.0003 * (b + 1) * torch.cos(200 * x[b, :, 0])
    sim.x = x
    sensor.update(0.01, force_recompute=True)
    flow = sensor.data.output["marker_motion"].cpu().numpy()  # (B,2,128,2)
    assert flow.shape == (B, 2, 128, 2)
    # oracle
    surf_ids = ms.surf_vertex_ids
    cam = lambda v: v - np.array([0.0, 0.0, -0.024])
    grid = gen_marker_grid()
    tri, wgt = gen_marker_weight(grid, cam(P[surf_ids]), ms.surf_triangles)
    init_uv = marker_uv(cam(P[surf_ids])[None].repeat(B, 0), tri, wgt)
    curr_uv = marker_uv(cam(x.cpu().numpy()[:, surf_ids]), tri, wgt)
    u0, v0 = init_uv[0, :, 0], init_uv[0, :, 1]
    mask = (u0 > 5) & (u0 < 240) & (v0 > 5) & (v0 < 320)  # (sic) u vs height, v vs width (VT:382-387)
    n = int(mask.sum())
    assert 0 < n < 128
    ref = np.zeros((B, 2, 128, 2))
    ref[:, 0, :n] = init_uv[:, mask]
    ref[:, 1, :n] = curr_uv[:, mask]
    ref[:, :, n:] = ref[:, :, n - 1:n]  # padded by repeating the last marker
    np.testing.assert_allclose(flow, ref, rtol=0, atol=2e-3)  # float32 output buffer of ~300 px values
    assert np.abs(flow[:, 1] - flow[:, 0]).max() > 1.0  # markers really moved
    assert np.abs(flow[0] - flow[2]).max() > 0.5         # per-env flows differ (the reference fills env 0 only)


def test_marker_flow_in_one_launch_vs_oracle_and_vs_the_general_path():
    """`tacex_fem_marker_flow` (static marker grid, more in-image markers than num_markers: VT:354-413 with the random subset of VT:394-399):
    the plugin's float32 output and the float64 flow against the NumPy oracle with the same RandomState draws, pixels and normalised; the
    projections of all markers (`curr_marker_uv`) against the oracle; and the general (multi-launch) path gives the same numbers."""
    from oracle.fem_oracle import marker_uv
    from tacex_amd.simulation_approaches.fem_based.sim.tactile_sensor_uipc import VisionTactileSensorUIPC, gen_marker_grid, gen_marker_weight
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    B, K = 3, 40
    P, Tt = gelpad_box_mesh(10, 8, 3, size=(0.030, 0.018, 0.0045))
    P = P - np.array([0.011, 0.009, 0.0])
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
    gel = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=Tt), sim)
    sim.setup_sim()
    cam_pos = torch.tensor([0.0005, -0.0003, -0.024], dtype=torch.float64)
    quat = torch.tensor([0.9990482, 0.0, 0.0, 0.0436194], dtype=torch.float64)  # 5 degrees about the optical axis (w, x, y, z)
    mk = lambda norm, seed: VisionTactileSensorUIPC(gel, sim, cam_pos, quat, num_markers=K, normalize=norm, seed=seed)
    x = sim.x.clone()
    for b in range(B):
        x[b, :, 0] += 0.0004 * (b + 1) * torch.sin(300 * x[b, :, 1])
        x[b, :, 2] += 0.0003 * (b + 1) * torch.cos(200 * x[b, :, 0])
    made = {(norm, role): mk(norm, 3) for norm in (False, True) for role in ("fused", "general", "again")}  # (reference surface and marker
    ms = made[(False, "fused")]                                                                              #  weights: the rest shape, at construction)
    many = VisionTactileSensorUIPC(gel, sim, cam_pos, quat, num_markers=4096)
    sim.x = x
    # oracle: camera frame = R^T (x - pos), R from the quaternion
    w_, x_, y_, z_ = quat.numpy()
    R = np.array([[1 - 2 * (y_**2 + z_**2), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_)],
                  [2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_**2 + z_**2), 2 * (y_ * z_ - x_ * w_)],
                  [2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_**2 + y_**2)]])
    cam = lambda v: (v - cam_pos.numpy()) @ R
    surf = ms.surf_vertex_ids
    tri, wgt = gen_marker_weight(gen_marker_grid(), cam(P[surf]), ms.surf_triangles)
    init_uv = marker_uv(cam(P[surf])[None].repeat(B, 0), tri, wgt)
    curr_uv = marker_uv(cam(x.cpu().numpy()[:, surf]), tri, wgt)
    mask = (init_uv[0, :, 0] > 5) & (init_uv[0, :, 0] < 240) & (init_uv[0, :, 1] > 5) & (init_uv[0, :, 1] < 320)
    ids = np.where(mask)[0]
    assert ids.size > K
    for norm in (False, True):
        ms = made[(norm, "fused")]
        rng = np.random.RandomState(3)
        gen_marker_grid(rng=rng)  # (the grid is drawn first, VT:189-247: a static grid consumes the same draws as a random one)
        out32 = torch.zeros((B, 2, K, 2), dtype=torch.float32, device="cuda:0")
        for call in range(2):  # two steps: two draws
            chosen = ids[rng.choice(ids.size, K, replace=False)]
            ref = np.stack([init_uv[:, chosen], curr_uv[:, chosen]], 1)
            if norm:
                ref = ref / 160.0 - 1.0
            if call == 0:
                flow = ms.gen_marker_flow()
                assert flow.dtype == torch.float64
                np.testing.assert_allclose(flow.cpu().numpy(), ref, rtol=1e-11, atol=1e-11)
            else:
                assert ms.gen_marker_flow_fused(out_f32=out32) is out32
                np.testing.assert_allclose(out32.cpu().numpy(), ref, rtol=0, atol=3e-5 if not norm else 2e-7)
            np.testing.assert_allclose(ms.curr_marker_uv.cpu().numpy(), curr_uv, rtol=1e-11)
        # the general path with the same draws
        gen = made[(norm, "general")]
        tri_d, wgt_d = gen._setup()
        a = gen._gen_marker_flow_static(tri_d, wgt_d, gen._project(gen.get_surface_vertices_camera(), tri_d, wgt_d))
        b_ = made[(norm, "again")]
        np.testing.assert_allclose(a.cpu().numpy(), b_.gen_marker_flow().cpu().numpy(), rtol=1e-12, atol=1e-12)
    # fewer in-image markers than asked for: all of them, padded by repeating the last (VT:400-405) - the same launch with a fixed list
    f = many.gen_marker_flow_fused().cpu().numpy()
    n = ids.size
    assert f.shape == (B, 2, 4096, 2)
    np.testing.assert_allclose(f[:, 0, :n], init_uv[:, ids], rtol=1e-11)
    np.testing.assert_allclose(f[:, 1, :n], curr_uv[:, ids], rtol=1e-11)
    assert (f[:, :, n:] == f[:, :, n - 1:n]).all()


@pytest.mark.parametrize("nu", [0.3, 0.49])
def test_hip_element_gradient_uniaxial_closed_form(nu):
    """The HIP element kernel against a CLOSED FORM (not the oracle): F = diag(s,1,1) on the unit right tet gives
    Pxx = mu (1 - 1/(s^2+3)) s + lambda (s - alpha), Pyy = Pzz = mu (1 - 1/(s^2+3)) + lambda (s - alpha) s with
    (mu, lambda) = (4/3 mu_L, lambda_L + 5/6 mu_L) of `youngs_poisson(E, nu)`; its slope at s = 1 is Hooke's law."""
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg

    E_mpa = 0.01
    E = E_mpa * 1e6
    mu_l, lam_l = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    mu, lam = 4.0 / 3.0 * mu_l, lam_l + 5.0 / 6.0 * mu_l
    alpha = 1.0 + 0.75 * mu / lam
    X = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    T = np.array([[0, 1, 2, 3]], dtype=np.int32)
    stretches = [0.7, 1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.3]
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=len(stretches))
    cfg = UipcObjectCfg(mesh_points=X, mesh_tets=T)
    cfg.constitution_cfg.youngs_modulus, cfg.constitution_cfg.poisson_rate = E_mpa, nu
    UipcObject(cfg, sim)
    sim.setup_sim()
    x = torch.from_numpy(np.stack([X * np.array([s, 1.0, 1.0]) for s in stretches])).cuda()
    _, g, _ = sim.element_terms(x, energy=False, hessian=False)
    g = g.cpu().numpy()[:, :, 0].reshape(len(stretches), 4, 3)  # (env, vertex, xyz)
    P = 6.0 * np.stack([g[:, 1], g[:, 2], g[:, 3]], -1)        # Dm = I, vol = 1/6
    for k, s in enumerate(stretches):
        pxx = mu * (1 - 1 / (s * s + 3)) * s + lam * (s - alpha)
        pyy = mu * (1 - 1 / (s * s + 3)) + lam * (s - alpha) * s
        np.testing.assert_allclose(np.diag(P[k]), [pxx, pyy, pyy], rtol=1e-11, atol=1e-9 * lam_l)
        assert np.abs(P[k] - np.diag(np.diag(P[k]))).max() <= 1e-9 * lam_l
    dP = (P[3] - P[1]) / 2e-6
    assert abs(dP[0, 0] - (lam_l + 2 * mu_l)) <= 1e-5 * (lam_l + 2 * mu_l)  # Hooke: lambda_L + 2 mu_L
    assert abs(dP[1, 1] - lam_l) <= 1e-5 * (lam_l + 2 * mu_l)              # Hooke: lambda_L
    assert np.abs(P[2]).max() <= 1e-9 * lam_l                                # rest stability at nu = 0.49


def test_attachment_chain_aim_set_constraints_step_vs_oracle():
    """a20 end to end for B envs with DISTINCT body poses: UipcIsaacAttachments.apply (compute_aim_positions + the animator's
    is_constrained / aim_position writes, UA:364-428) -> UipcSim.step, against the oracle driven the same way."""
    from oracle.fem_oracle import FemModel, attachment_aim_positions, box_tet_mesh
    from tacex_amd.uipc import UipcIsaacAttachments, UipcIsaacAttachmentsCfg, UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg

    P, Tt = box_tet_mesh(4, 5, 2)
    B = 3
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
    sim.cfg.linear_system.coarse_grid = None  # compared with the oracle's block-Jacobi step
    sim.cfg.linear_system.vertex_chains = None
    gel = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=Tt), sim)
    sim.setup_sim(constraint_strength_ratio=100.0)
    sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 200, 1e-8
    # the "sensor case": a box collider hugging the back face (z = 0) of the gelpad, body frame at its centre
    size = P.max(0) - P.min(0)
    body_pos0 = np.array([size[0] / 2, size[1] / 2, -0.001])
    att = UipcIsaacAttachments(UipcIsaacAttachmentsCfg(), gel, rigid_collider=("box", (size[0] / 2 + 1e-6, size[1] / 2 + 1e-6, 0.001)),
                               rigid_pos=body_pos0)
    back = np.where(P[:, 2] < 1e-12)[0]
    assert sorted(att.attachment_points_idx.tolist()) == sorted(back.tolist())  # exactly the back-face vertices are attached
    np.testing.assert_allclose(att.attachment_offsets, (P[att.attachment_points_idx] - body_pos0).astype(np.float32), atol=1e-9)
    # distinct poses: translation + yaw / roll per env
    ang = np.array([0.0, 0.05, -0.08])
    quat = np.stack([np.cos(ang / 2), np.sin(ang / 2) * np.array([0, 0, 1.0]), np.zeros(3), np.sin(ang / 2) * np.array([0, 1.0, 0])], -1)
    quat = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    pos = body_pos0[None] + np.array([[0, 0, 0], [0.0003, 0, 0.0002], [-0.0002, 0.0004, -0.0003]])
    aim = att.apply(sim, torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda())
    aim_o = attachment_aim_positions(att.attachment_offsets, pos, quat)
    assert np.abs(aim.cpu().numpy() - aim_o).max() <= 2e-7 * np.abs(aim_o).max()  # float32 rotation on both sides
    cons = sim.is_constrained.cpu().numpy()
    assert (cons[:, att.attachment_points_idx] == 1).all() and cons.sum() == B * len(back)
    np.testing.assert_array_equal(sim.aim_position[:, att.attachment_points_idx].cpu().numpy(), aim.cpu().numpy())
    # one backward-Euler step (3 Newton iterations) on the GPU and in the oracle
    m = FemModel.build(P, Tt, youngs=1e4, strength=100.0)
    g = np.array(sim.cfg.gravity)
    sim.newton_kwargs = None
    x_n = sim.x.clone()
    sim.x_tilde = x_n + sim.cfg.dt * sim.v + sim.cfg.dt**2 * sim._g
    for _ in range(3):
        sim.newton_step()
    x = sim.x.cpu().numpy()
    c = np.zeros(len(P)); c[att.attachment_points_idx] = 1.0
    xt = P + m.dt**2 * g
    for b in range(B):
        aim_b = P.copy()
        aim_b[att.attachment_points_idx] = aim.cpu().numpy()[b]
        xo = P.copy()
        for _ in range(3):
            xo, _ = m.newton_step(xo, xt, c, aim_b, pcg_max_iter=200, pcg_tol_rate=1e-8)
        assert np.abs(x[b] - xo).max() <= 1e-6 * np.ptp(P), b
    # and through the public step(): the attached face follows its body, envs differ
    sim.x.copy_(x_n)
    sim.step(max_newton_iter=6)
    d = (sim.x[:, att.attachment_points_idx] - aim).abs().amax().item()
    assert d < 2e-4
    assert (sim.x[1] - sim.x[2]).abs().max().item() > 1e-4


def _contact_setup(B=3, strength=100.0):
    """A gelpad-sized block whose back face (z = 0) is glued; per env a different indenter hovering over / touching the front."""
    from oracle.fem_oracle import FemModel, box_tet_mesh
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg

    P, Tt = box_tet_mesh(4, 5, 2)
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
    sim.cfg.linear_system.coarse_grid = None  # block-Jacobi path (the C4 tests run the two-level preconditioner)
    sim.cfg.linear_system.vertex_chains = None
    gel = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=Tt), sim)
    sim.setup_sim(constraint_strength_ratio=strength)
    back = np.where(P[:, 2] < 1e-12)[0]
    sim.set_constraints(back, torch.from_numpy(P[back]).cuda()[None].repeat(B, 1, 1))
    top = P[:, 2].max()
    fr = np.where(P[:, 2] > top - 1e-12)[0]
    vc = fr[np.argmin(np.hypot(P[fr, 0] - P[:, 0].mean(), P[fr, 1] - P[:, 1].mean()))]  # the barrier is per VERTEX: aim at one
    cx, cy = P[vc, 0], P[vc, 1]
    ind = np.zeros((B, 8))
    ind[0] = [1, cx, cy, top + 0.004 + 0.0004, 0.004, 0, 0, 1]              # sphere, lowest point 0.4 mm above the pad: inside d_hat
    ind[1] = [2, cx, cy, top + 0.0006, 0, 0, 0, -1.0]                        # half-space coming down from above (solid side z > c)
    if B > 2:
        ind[2] = [0, 0, 0, 0, 0, 0, 0, 0]                                    # no indenter
    if B > 3:  # lying capsule (a pin across the pad): radius 3 mm, half axis 8 mm along x, lowest line 0.5 mm above the pad
        ind[3] = [3, cx, cy, top + 0.003 + 0.0005, 0.003, 0.008, 0.0, 0.0]
    m = FemModel.build(P, Tt, youngs=1e4, strength=strength)
    return sim, gel, m, P, back, ind


def test_contact_energy_gradient_vs_oracle():
    from oracle.fem_oracle import ContactModel

    sim, gel, m, P, back, ind = _contact_setup(B=4)
    B = sim.num_envs
    rng = np.random.default_rng(2)
    x = P[None] + 2e-5 * rng.normal(size=(B,) + P.shape)
    sim.x = torch.from_numpy(x).cuda()
    sim.x_tilde = sim.x.clone()
    E0 = sim.energy().cpu().numpy()
    g0 = sim.gradient().cpu().numpy()
    sim.set_contact_indenters(torch.from_numpy(ind))
    E1 = sim.energy().cpu().numpy()
    g1 = sim.gradient().cpu().numpy()
    area = gel.surface_vertex_areas()
    np.testing.assert_allclose(area.sum(), 2 * (20.75 * 25.25 + 20.75 * 4.5 + 25.25 * 4.5) * 1e-6, rtol=1e-12)  # the block's surface
    kappa = sim.cfg.contact.default_contact_resistance * 1e9 * sim.cfg.contact.d_hat
    for b in range(B):
        cm = ContactModel(area, ind[b], sim.cfg.contact.d_hat, kappa, sim.cfg.dt)
        ec, gc = cm.energy(x[b]), cm.gradient(x[b])
        assert abs((E1[b] - E0[b]) - ec) <= 1e-10 * max(abs(ec), 1e-30) + 1e-12 * abs(E0[b]), b
        assert np.abs((g1[b] - g0[b]) - gc).max() <= 1e-10 * max(np.abs(gc).max(), 1e-30) + 1e-12 * np.abs(g0[b]).max(), b
    assert E1[0] > E0[0] and E1[1] > E0[1] and E1[2] == E0[2]  # envs 0 / 1 are inside d_hat, env 2 has no indenter
    assert E1[3] > E0[3] and (sim.contact_gaps()[3] < sim.cfg.contact.d_hat).sum().item() >= 3  # the capsule touches a line of vertices
    # penetration = infinite energy
    xp = x.copy(); xp[1, :, 2] += 0.001
    sim.x = torch.from_numpy(xp).cuda()
    assert np.isinf(sim.energy().cpu().numpy()[1])
    sim.set_contact_indenters(None)
    np.testing.assert_array_equal(sim.energy(torch.from_numpy(x).cuda()).cpu().numpy(), E0)


def test_contact_newton_step_vs_oracle_and_no_penetration():
    """Indenters are pushed INTO the pad step by step (each move smaller than the current gap, as a CCD-filtered rigid motion
    would be): Newton iterations with the barrier + CCD step filter follow the oracle, the energy never increases, and no
    surface vertex ever crosses the indenter surface although the indenter ends up 0.8 mm below the undeformed front face."""
    from oracle.fem_oracle import ContactModel, newton_step_contact

    sim, gel, m, P, back, ind = _contact_setup(B=2)
    sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 300, 1e-10
    area = gel.surface_vertex_areas()
    dhat = sim.cfg.contact.d_hat
    kappa = sim.cfg.contact.default_contact_resistance * 1e9 * dhat
    cons = np.zeros(len(P)); cons[back] = 1.0
    xo = [P.copy(), P.copy()]
    sim.x_tilde = sim.x.clone()
    top = P[:, 2].max()
    for move in range(6):
        gaps = sim.contact_gaps().amin(1).cpu().numpy() if move else np.array([4e-4, 6e-4])
        ind[0, 3] -= 0.5 * min(gaps[0], 4e-4); ind[1, 3] -= 0.5 * min(gaps[1], 4e-4)   # never more than half the gap
        sim.set_contact_indenters(torch.from_numpy(ind))
        for it in range(4):
            st = sim.newton_step().cpu().numpy().copy()
            assert (st[:, 1] <= st[:, 0] * (1 + 1e-12) + 1e-18).all(), (move, it, st)
            assert float(sim.contact_gaps().amin()) > 0.0, "a vertex crossed the indenter surface"
            for b in range(2):
                cm = ContactModel(area, ind[b], dhat, kappa, sim.cfg.dt)
                xo[b], so = newton_step_contact(m, cm, xo[b], P, cons, P, pcg_max_iter=300, pcg_tol_rate=1e-10)
                assert abs(st[b, 0] - so[0]) <= 1e-7 * abs(so[0]) + 1e-18, (move, it, b, st[b], so)
                assert abs(st[b, 1] - so[1]) <= 1e-5 * abs(so[1]) + 1e-18, (move, it, b, st[b], so)
    x = sim.x.cpu().numpy()
    for b in range(2):
        assert np.abs(x[b] - xo[b]).max() <= 1e-5 * np.ptp(P), b
    # the front face is really dented, and only in front of the indenter for the sphere
    dent = top - x[:, :, 2]
    front = P[:, 2] > top - 1e-9
    assert dent[0][front].max() > 1e-4 and dent[1][front].min() > 1e-4
    r = np.hypot(P[front, 0] - ind[0, 1], P[front, 1] - ind[0, 2])
    assert dent[0][front][r > 0.008].max() < 0.3 * dent[0][front].max()


# ---- BASELINE config 4 size: the 8 x 10 x 4 gelpad (495 vertices / 1 920 tets) the LDS window / CSR cursor logic is sized for --------
def _lag(sim):
    """The oracle's name for the friction lag the sim's cfg selects ("ipc" = the default since round 6; "capped" = the oracle's "start")."""
    return {"ipc": "ipc", "capped": "start"}[sim.cfg.contact.friction_lag]


def _chains(sim):
    """(next, heads) of the chains the library was given, for the oracle."""
    from oracle.fem_oracle import chain_tables

    return chain_tables(sim.vertex_chains, sim._obj.num_verts)


def _c4_scene(B, strength=1000.0):
    """The bench's gelpad (tacex_amd/uipc/gelpad_scene.py) restated for the oracle: back face constrained (sheared a little), a
    sphere over the middle of the front face just inside d_hat."""
    from oracle.fem_oracle import ContactModel, FemModel
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    P, T = gelpad_box_mesh(8, 10, 4)
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
    gel = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
    sim.setup_sim(constraint_strength_ratio=strength)
    m = FemModel.build(P, T, youngs=gel.cfg.constitution_cfg.youngs_modulus * 1e6, poisson=gel.cfg.constitution_cfg.poisson_rate,
                       density=gel.cfg.mass_density, dt=sim.cfg.dt, strength=strength)
    back = np.where(P[:, 2] < 1e-12)[0]
    aim = np.repeat(P[None], B, 0)
    aim[:, :, 0] += 0.0002 * (1 + np.arange(B))[:, None]  # every env sheared differently
    sim.set_constraints(back, torch.from_numpy(aim[:, back]).cuda())
    cons = np.zeros(len(P)); cons[back] = 1.0
    top, size = P[:, 2].max(), P.max(0)
    ind = np.zeros((B, 8)); ind[:, 0] = 1.0
    ind[:, 1], ind[:, 2], ind[:, 4] = size[0] / 2, size[1] / 2, 0.004
    ind[:, 3] = top + 0.004 + 0.0009 - 1e-4 * np.arange(B)  # gaps of 0.9, 0.8, ... mm: inside d_hat = 1 mm
    sim.set_contact_indenters(torch.from_numpy(ind))
    area = gel.surface_vertex_areas()
    kappa = sim.cfg.contact.default_contact_resistance * 1e9 * sim.cfg.contact.d_hat
    cms = [ContactModel(area, ind[b], sim.cfg.contact.d_hat, kappa, sim.cfg.dt) for b in range(B)]
    return sim, m, P, cons, aim, cms


def test_newton_step_c4_mesh_vs_oracle():
    """Two Newton iterations on the C4 gelpad (four 512-tet LDS windows, 495 of 512 vertex threads), attachments + sphere contact,
    against the oracle: energies before / after, step length, PCG iteration count and positions."""
    from oracle.fem_oracle import newton_step_contact

    B = 2
    sim, m, P, cons, aim, cms = _c4_scene(B)
    sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-12
    sim.x_tilde = sim.x.clone()
    xo = [P.copy() for _ in range(B)]
    for it in range(2):
        st = sim.newton_step().cpu().numpy().copy()
        for b in range(B):
            xo[b], so = newton_step_contact(m, cms[b], xo[b], P, cons, aim[b], pcg_max_iter=600, pcg_tol_rate=1e-12, coarse=sim.coarse_space, chains=_chains(sim))
            assert abs(st[b, 0] - so[0]) <= 1e-6 * abs(so[0]) + 1e-20, (it, b, st[b], so)
            assert abs(st[b, 1] - so[1]) <= 1e-5 * abs(so[1]) + 1e-20, (it, b, st[b], so)
            assert st[b, 2] == so[2], (it, b, st[b], so)
            assert abs(st[b, 3] - so[3]) <= 0.05 * so[3] + 2, (it, b, st[b], so)  # PCG iterations (summation order differs)
            assert so[3] <= 40, so  # the two-level preconditioner at work (block Jacobi alone: > 100 at this tolerance)
    x = sim.x.cpu().numpy()
    for b in range(B):
        assert np.abs(x[b] - xo[b]).max() <= 1e-6 * np.ptp(P), b
    assert np.abs(x[0] - x[1]).max() > 1e-5  # the envs really differ


def test_step_c4_vs_oracle_step_and_convergence_rule():
    """UipcSim.step() = ONE tacex_fem_step call (predictor, in-kernel Newton loop with device-side exit, velocity) against the
    oracle's fem_step over three time steps with a moving indenter: same iteration counts, positions, velocities.  The
    convergence rule: the test looks at the UNSCALED Newton direction - here the CCD filter shortens the first
    iterations, which must not count as converged although their update is tiny."""
    from oracle.fem_oracle import fem_step

    B = 2
    sim, m, P, cons, aim, cms = _c4_scene(B)
    sim.cfg.newton.velocity_tol = 2e-3  # [m/s]: 20 um per step - tight enough to need several iterations
    sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-12
    xo = [P.copy() for _ in range(B)]
    vo = [np.zeros_like(P) for _ in range(B)]
    ind = sim.contact_indenters
    saw_truncated = False
    fric = (sim.cfg.contact.default_friction_ratio, sim.cfg.contact.eps_velocity)
    prev = ind[:, 1:4].cpu().numpy().copy()
    for k in range(3):
        ind[:, 1] += 2e-5       # the indenter slides sideways: friction (on by default, uipc_sim.py:103-124) drags the surface along
        gap = sim.contact_gaps().amin(1)
        ind[:, 3] -= 0.15 * gap  # ... and approaches by less than the gap (the documented contract)
        cur = ind[:, 1:4].cpu().numpy().copy()
        disp = cur - prev if k > 0 else np.zeros_like(cur)  # the first step after set_contact_indenters sees no indenter motion
        prev = cur
        for b in range(B):
            cms[b].ind[1:4] = cur[b]
        sim.step(max_newton_iter=24)
        info = sim.check_step()
        assert len(info["penetrating_envs"]) == 0 and len(info["line_search_failed_envs"]) == 0
        assert info["newton_iters"].max() < 24  # converged, not capped
        x, v = sim.x.cpu().numpy(), sim.v.cpu().numpy()
        for b in range(B):
            xo[b], vo[b], io = fem_step(m, cms[b], xo[b], vo[b], cons, aim[b], gravity=sim.cfg.gravity, max_newton=24,
                                        velocity_tol=2e-3, pcg_max_iter=600, pcg_tol_rate=1e-12, coarse=sim.coarse_space, chains=_chains(sim),
                                        friction=(fric[0], fric[1], disp[b]), friction_lag=_lag(sim))
            # same iteration count (a convergence test that falls within round-off of its threshold may differ by one iteration);
            # both stop inside the Newton tolerance of 20 um and their PCG round-off differs by ~0.1 um
            assert abs(int(info["newton_iters"][b]) - int(io[0])) <= 1, (k, b, info["newton_iters"], io)
            tol = 1e-4 * np.ptp(P) if info["newton_iters"][b] == io[0] else 2 * 2e-3 * sim.cfg.dt
            assert np.abs(x[b] - xo[b]).max() <= tol, (k, b)
            assert np.abs(v[b] - vo[b]).max() <= tol / sim.cfg.dt, (k, b)
            saw_truncated |= io[0] > 1
        assert float(sim.contact_gaps().amin()) > 0.0
    assert saw_truncated
    # a penetrating indenter is reported, not swallowed
    ind[0, 3] -= 2.0 * float(sim.contact_gaps()[0].amin())
    sim.step(max_newton_iter=2)
    with pytest.raises(RuntimeError, match="penetrated"):
        sim.check_step()
    assert 0 in sim.check_step(raise_on_penetration=False)["penetrating_envs"]


@pytest.mark.parametrize("res", [(240, 320), (480, 640)])  # C4 (320x240) and C5 (640x480: BASELINE configs[4]) through the same path
def test_fem_gelpad_scene_through_the_sensor(res):
    """The C4 / C5 scene of bench.py (tacex_amd.uipc.gelpad_scene.FemGelpad) stepped for 8 envs and read through
    GelSightSensor.update() with the FEM-driven marker plugin: finite state, no penetration, markers move, envs differ."""
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.simulation_approaches.fem_based import ManiSkillSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg
    from tacex_amd.uipc.gelpad_scene import FemGelpad
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    B, (H, W) = 8, res
    fem = FemGelpad(B, "cuda:0")
    assert fem.num_tets == 1920 and fem.num_verts == 495
    cfg = GelSightSensorCfg(
        num_envs=B, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
        data_types=["tactile_rgb", "height_map", "marker_motion"],
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(CALIB_GELSIGHT_MINI), gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,
                                          with_shadow=False, tactile_img_res=(W, H), device="cuda:0"),
        marker_motion_sim_cfg=ManiSkillSimulatorCfg(tactile_img_res=(W, H), device="cuda:0", camera_pos_w=(0.008, 0.012625, -0.024)),
        device="cuda:0")
    s = GelSightSensor(cfg, gelpad_obj=fem.gelpad)
    s.initialize()
    hm, _ = synthetic_depth_maps(B, H, W, seed=5)
    s.set_camera_depth((hm / 1000.0).cuda())
    first = None
    for i in range(12):
        fem.step(i)
        s.update(dt=0.01, force_recompute=True)
        md = s.data.output["marker_motion"]
        if first is None:
            first = md.clone()
    info = fem.sim.check_step()
    assert len(info["penetrating_envs"]) == 0
    x = fem.sim.x
    assert torch.isfinite(x).all() and torch.isfinite(s.data.output["tactile_rgb"]).all() and torch.isfinite(md).all()
    assert float(fem.sim.contact_gaps().amin()) > 0.0
    P = torch.from_numpy(fem.gelpad.points).cuda()
    dent = (P[None, :, 2] - x[:, :, 2]).amax(1)
    assert float(dent.min()) > 5e-5 and float((dent.max() - dent.min())) > 1e-5  # every pad is dented, by different amounts (depth ramp)
    assert float((md - first).abs().max()) > 0.05 * (W / 320)   # markers moved [px]
    assert float((md[0] - md[-1]).abs().max()) > 1e-3  # the envs differ


def test_fem_step_on_a_side_stream_overlaps_the_sensor_update_and_changes_nothing():
    """`FemGelpad(side_stream=True)` (what bench.py's C4 / C5 entries run): scene driver + FEM step on a HIP stream of their own, the
    sensor update on the caller's stream, the FEM-driven marker plugin waiting for `UipcSim.step_done`.  Same markers, frames and pad
    state as the single-stream run, step by step (deterministic sweeps, so that the comparison is bit for bit), with a busy kernel
    queue on the caller's stream in between - without the event the marker kernel would read the pad mid-step."""
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.simulation_approaches.fem_based import ManiSkillSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg
    from tacex_amd.uipc.gelpad_scene import FemGelpad
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    B, H, W = 8, 240, 320
    outs = {}
    for side in (False, True):
        fem = FemGelpad(B, "cuda:0", max_newton_iter=40, side_stream=side)
        assert fem.sim._lib.tacex_fem_set_deterministic(fem.sim._handle, 1) == 0
        cfg = GelSightSensorCfg(
            num_envs=B, sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
            data_types=["tactile_rgb", "height_map", "marker_motion"],
            optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(CALIB_GELSIGHT_MINI), gelpad_height=0.0045, gelpad_to_camera_min_distance=0.024,
                                              with_shadow=False, tactile_img_res=(W, H), device="cuda:0"),
            marker_motion_sim_cfg=ManiSkillSimulatorCfg(tactile_img_res=(W, H), device="cuda:0", camera_pos_w=(0.008, 0.012625, -0.024)),
            device="cuda:0")
        s = GelSightSensor(cfg, gelpad_obj=fem.gelpad)
        s.initialize()
        hm, _ = synthetic_depth_maps(B, H, W, seed=5)
        s.set_camera_depth((hm / 1000.0).cuda())
        rec = []
        for i in range(10):
            fem.step(i)
            s.update(dt=0.01, force_recompute=True)
            rec.append((s.data.output["marker_motion"].clone(), s.data.output["tactile_rgb"].clone()))
        torch.cuda.synchronize()
        assert (fem.sim.step_done is not None) == side
        outs[side] = (rec, fem.sim.x.clone(), fem.sim.step_info.clone())
    for (m0, r0), (m1, r1) in zip(outs[False][0], outs[True][0]):
        assert torch.equal(m0, m1) and torch.equal(r0, r1)
    assert torch.equal(outs[False][1], outs[True][1]) and torch.equal(outs[False][2], outs[True][2])
    assert float((outs[True][0][-1][0] - outs[True][0][0][0]).abs().max()) > 0.05  # the markers moved [px]


def test_friction_drags_the_pad_surface():
    """A sphere pressed into the pad slides sideways: with Coulomb friction (reference default, ratio 0.5) the contact patch of
    the surface follows it, without friction it stays; kernel and oracle agree on how far.  The press is gentle enough for every
    step to converge well inside the iteration cap (9-18 Newton iterations): pressed at 0.45 of the gap per step the solver runs
    into failing line searches and kernel and oracle part within the backtracking (profiles/r03_experiments.md section 10)."""
    from oracle.fem_oracle import fem_step

    res = {}
    slide = 5e-5
    for mu in (0.5, 0.0):
        sim, m, P, cons, aim, cms = _c4_scene(1)
        sim.cfg.contact.default_friction_ratio = mu
        sim.cfg.contact.enable_friction = mu > 0
        sim.cfg.newton.velocity_tol = 1e-3
        sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-12
        sim.set_contact_indenters(sim.contact_indenters)  # re-reads the friction settings
        ind = sim.contact_indenters
        xo, vo = P.copy(), np.zeros_like(P)
        prev = None
        for k in range(7):
            if k >= 3:
                ind[:, 1] += slide            # after pressing: slide along x, 50 um per step (less than the gap it leaves)
            gap = float(sim.contact_gaps().amin())
            if k < 3:
                ind[:, 3] -= 0.3 * gap        # press
            elif gap < 2 * slide:
                ind[:, 3] += 2 * slide - gap  # (keep the contract: never closer than the sideways step)
            cur = ind[0, 1:4].cpu().numpy().copy()
            disp = cur - prev if prev is not None else np.zeros(3)
            prev = cur
            cms[0].ind[1:4] = cur
            sim.step(max_newton_iter=60)
            info = sim.check_step()
            assert len(info["penetrating_envs"]) == 0 and len(info["line_search_failed_envs"]) == 0 and int(sim.last_newton_iters) < 60
            if mu > 0:
                xo, vo, io = fem_step(m, cms[0], xo, vo, cons, aim[0], gravity=sim.cfg.gravity, max_newton=60, velocity_tol=1e-3, pcg_max_iter=600,
                                      pcg_tol_rate=1e-12, coarse=sim.coarse_space, chains=_chains(sim), friction=(mu, sim.cfg.contact.eps_velocity, disp), friction_lag=_lag(sim))
                assert io[0] < 60 and int(io[2]) & 3 == 0
                assert np.abs(sim.x[0].cpu().numpy() - xo).max() <= 2 * 1e-3 * sim.cfg.dt, k  # both inside the Newton tolerance of the same state
        x = sim.x[0].cpu().numpy()
        top = P[:, 2] > P[:, 2].max() - 1e-9
        near = top & (np.hypot(P[:, 0] - cur[0], P[:, 1] - cur[1]) < 0.004)
        res[mu] = float((x[near, 0] - P[near, 0]).mean())
        if mu > 0:
            res[(mu, "oracle")] = float((xo[near, 0] - P[near, 0]).mean())
    print("drag [m]:", res)
    assert res[0.5] > 5e-5 and res[0.5] > 1.25 * abs(res[0.0]), res  # dragged along +x, beyond what the dent's slope alone pushes
    assert abs(res[0.5] - res[(0.5, "oracle")]) <= 0.01 * res[0.5], res  # (measured 1.5e-5 relative)


def test_mesh_indenter_vs_oracle_and_analytic_sphere():
    """Indenter kind 4 (n4, next slice): a rigid triangle mesh pressed into the pad.  (1) The torch diagnostic and the kernel's
    distance agree with the oracle's point-triangle distance on a tilted box (faces, edges, corners).  (2) Newton iterations
    against the oracle with an icosphere mesh: energies, step, PCG count, positions.  (3) Whole steps with the icosphere track the
    analytic sphere of the same radius (the mesh lies inside the sphere by the faces' sagitta)."""
    from oracle.fem_oracle import ContactModel, contact_distance, newton_step_contact
    from tacex_amd.uipc.indenter_meshes import box, icosphere

    B = 2
    sim, m, P, cons, aim, cms = _c4_scene(B)
    sim.cfg.contact.enable_friction = False
    top, size = P[:, 2].max(), P.max(0)
    area = cms[0].area
    # (1) distance diagnostic vs the oracle: a tilted, inflated box corner-down over the pad
    bv, bt = box((0.002, 0.003, 0.0015))
    sim.set_indenter_mesh(bv, bt)
    ind = np.zeros((B, 8)); ind[:, 0] = 4.0
    ind[:, 1], ind[:, 2], ind[:, 3], ind[:, 4] = size[0] / 2, size[1] / 2, top + 0.004, 2e-4
    ind[0, 5:8] = [0.4, -0.7, 0.2]
    ind[1, 5:8] = [-1.1, 0.3, 0.9]
    sim.set_contact_indenters(torch.from_numpy(ind))
    g = sim.contact_gaps().cpu().numpy()         # tacex_fem_contact_gaps: the solver's own distance function
    gt = sim.contact_gaps_torch().cpu().numpy()  # its torch restatement
    for b in range(B):
        d, _ = contact_distance(ind[b], P, (bv, bt))
        np.testing.assert_allclose(g[b], d, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(gt[b], d, rtol=1e-12, atol=1e-15)
    # (2) Newton iterations with an icosphere of the scene's sphere radius, against the oracle
    sv, st = icosphere(0.004, 2)
    sim.set_indenter_mesh(sv, st)
    ind[:, 4] = 0.0
    ind[:, 5:8] = [[0.2, 0.1, -0.3], [0.0, 0.0, 0.0]]  # the pose matters: the facets are not symmetric
    ind[:, 3] = top + 0.004 + np.array([0.0008, 0.0007])
    sim.set_contact_indenters(torch.from_numpy(ind))
    kappa = sim.cfg.contact.default_contact_resistance * 1e9 * sim.cfg.contact.d_hat
    cmm = [ContactModel(area, ind[b].copy(), sim.cfg.contact.d_hat, kappa, sim.cfg.dt, mesh=(sv, st)) for b in range(B)]
    sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-12
    sim.x_tilde = sim.x.clone()
    xo = [P.copy() for _ in range(B)]
    for it in range(2):
        stt = sim.newton_step().cpu().numpy().copy()
        x = sim.x.cpu().numpy()
        for b in range(B):
            xo[b], so = newton_step_contact(m, cmm[b], xo[b], P, cons, aim[b], pcg_max_iter=600, pcg_tol_rate=1e-12, coarse=sim.coarse_space, chains=_chains(sim))
            assert abs(stt[b, 0] - so[0]) <= 1e-6 * abs(so[0]) + 1e-18 and abs(stt[b, 1] - so[1]) <= 1e-5 * abs(so[1]) + 1e-18, (it, b, stt[b], so)
            assert stt[b, 2] == so[2] and abs(stt[b, 3] - so[3]) <= 2, (it, b, stt[b], so)
            assert np.abs(x[b] - xo[b]).max() <= 1e-6 * np.ptp(P), (it, b)
    assert (sim.contact_gaps().amin(1) < sim.cfg.contact.d_hat).all()  # the barrier really acted
    # (3) steps: the icosphere tracks the analytic sphere
    res = {}
    for kind in (4, 1):
        s2, _, _, _, _, _ = _c4_scene(1)
        s2.cfg.contact.enable_friction = False
        # (both runs solved well inside the 10 % the comparison allows: the default tolerances - 0.5 mm per step on the Newton direction,
        #  1e-3 on r.z in the PCG - are looser than the difference between a faceted and a smooth sphere)
        s2.cfg.newton.velocity_tol = 2e-3
        s2.cfg.linear_system.tol_rate = 1e-8
        if kind == 4:
            s2.set_indenter_mesh(*icosphere(0.004, 3))
        row = np.array([[float(kind), size[0] / 2, size[1] / 2, top + 0.004 + 0.0009, 0.004 if kind == 1 else 0.0, 0, 0, 0]])
        s2.set_contact_indenters(torch.from_numpy(row))
        i2 = s2.contact_indenters
        for k in range(6):
            i2[:, 3] -= 0.4 * float(s2.contact_gaps().amin())
            s2.step(max_newton_iter=30)
            assert len(s2.check_step()["penetrating_envs"]) == 0
        res[kind] = s2.x[0].cpu().numpy()
    dent = (P[:, 2] - res[1][:, 2]).max()
    assert dent > 1e-4                                                   # the pad is dented by > 0.1 mm ...
    assert np.abs(res[4] - res[1]).max() <= 0.1 * dent, (np.abs(res[4] - res[1]).max(), dent)  # ... and the mesh agrees within 10 % of it


def test_fem_table_setters_reject_bad_input_and_gaps_without_indenter():
    """Error behaviour of the round-3 setters (tacex_fem_set_chains / set_indenter_mesh / set_coarse_space) and the gap query."""
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    P, T = gelpad_box_mesh(3, 3, 2)
    sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=2)
    UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
    sim.setup_sim()
    lib, h = sim._lib, sim._handle
    assert torch.isinf(sim.contact_gaps()).all()  # no indenter yet
    off = np.array([0, 2, 4], np.int32)
    vtx = np.array([0, 1, 1, 2], np.int32)  # vertex 1 in two chains
    assert lib.tacex_fem_set_chains(h, 2, off.ctypes.data, vtx.ctypes.data) != 0
    assert b"two chains" in lib.tacex_last_error()
    vtx = np.array([0, 1, 2, len(P)], np.int32)  # out of range
    assert lib.tacex_fem_set_chains(h, 2, off.ctypes.data, vtx.ctypes.data) != 0
    assert lib.tacex_fem_set_chains(h, 0, 0, 0) == 0
    v = np.zeros((3, 3)); t = np.array([[0, 1, 3]], np.int32)  # triangle vertex out of range
    assert lib.tacex_fem_set_indenter_mesh(h, 3, v.ctypes.data, 1, t.ctypes.data) != 0
    assert lib.tacex_fem_set_indenter_mesh(h, 0, 0, 0, 0) == 0
    assert lib.tacex_fem_set_coarse_space(h, 65, 0, 0, 0) != 0  # more than 64 coarse nodes
    # a kind-4 row without a mesh is no indenter: +inf gaps, the step runs
    ind = torch.zeros((2, 8), dtype=torch.float64); ind[:, 0] = 4.0
    sim.set_contact_indenters(ind)
    assert torch.isinf(sim.contact_gaps()).all()
    sim.step(max_newton_iter=2)
    assert torch.isfinite(sim.x).all()


def test_animated_aims_do_not_rebuild_the_preconditioner_and_setters_free_their_tables():
    """`set_constraints` called every step with the SAME vertex set (a caller animating the aim positions) must not
    re-run `refresh_preconditioner` (device sync, dense inverse, six table uploads); a NEW vertex does; and the C setters free
    the tables they replace - 40 forced refreshes leave the device memory where it was."""
    sim, m, P, cons, aim, cms = _c4_scene(2)
    sim.step(max_newton_iter=2)
    assert not sim._precond_dirty
    idx = np.nonzero(cons)[0]
    aims = sim.aim_position[:, idx].clone()
    calls = []
    orig = sim.refresh_preconditioner
    sim.refresh_preconditioner = lambda: (calls.append(1), orig())[1]
    idx_dev = torch.as_tensor(idx, device=sim.device)
    for k in range(5):
        sim.set_constraints(idx.tolist() if k % 2 else idx_dev, aims + 1e-6 * k)
        sim.step(max_newton_iter=2)
    assert calls == [] and not sim._precond_dirty
    free = [v for v in range(len(P)) if not cons[v]][:1]
    sim.set_constraints(free, sim.x[:, free].clone())
    assert sim._precond_dirty
    sim.step(max_newton_iter=2)
    assert calls == [1]
    torch.cuda.synchronize()
    sv, st = __import__("tacex_amd.uipc.indenter_meshes", fromlist=["icosphere"]).icosphere(0.004, 2)
    for k in range(3):  # reach the allocator's steady state first
        orig(); sim.set_indenter_mesh(sv, st)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(40):
        orig()
        sim.set_indenter_mesh(sv, st)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= (2 << 20), (free0, torch.cuda.mem_get_info()[0])  # was ~0.4 MB per refresh
    sim.step(max_newton_iter=2)  # the replaced tables are the ones in use
    assert torch.isfinite(sim.x).all()


def test_friction_sees_indenter_motion_through_a_new_tensor_every_step():
    """a caller who moves the indenter by handing `set_contact_indenters` a NEW tensor each step used to get zero
    indenter displacement (the call reset the previous positions) and friction against the world frame.  Now the two ways of moving
    the indenter - in-place mutation and a fresh tensor per step - give bit-identical trajectories."""
    res = []
    for fresh in (False, True):
        sim, m, P, cons, aim, cms = _c4_scene(1)
        sim.cfg.newton.velocity_tol = 1e-3
        assert sim._lib.tacex_fem_set_deterministic(sim._handle, 1) == 0  # fixed summation order: the comparison below is bit for bit
        sim.set_contact_indenters(sim.contact_indenters)
        row = sim.contact_indenters.clone()
        for k in range(6):
            gap = float(sim.contact_gaps().amin())
            if k < 3:
                row[:, 3] -= 0.3 * gap
            else:
                row[:, 1] += 5e-5
                if gap < 1e-4:
                    row[:, 3] += 1e-4 - gap
            if fresh:
                sim.set_contact_indenters(row.clone())
            else:
                sim.contact_indenters.copy_(row)
            sim.step(max_newton_iter=60)
        res.append(sim.x.cpu().numpy().copy())
    top = P[:, 2] > P[:, 2].max() - 1e-9
    assert (res[0][0][top, 0] - P[top, 0]).max() > 2e-5  # friction dragged the surface along +x
    np.testing.assert_array_equal(res[0], res[1])  # (deterministic sweeps, see above: the two ways of moving the indenter are the SAME computation)


def test_contact_following_start_is_only_an_initial_guess_and_tames_the_retreat():
    """The contact-following start of the Newton loop (`UipcSimCfg.contact.follow_indenter`, fem_newton_lds_kernel): vertices in the
    barrier zone start a step displaced with the indenter.  With the indenter RETREATING the followed loop needs a fraction of the
    Newton / PCG iterations of the loop started from the current positions (libuipc's start), and both end in the same state to the
    Newton tolerance; kernel and oracle agree with it switched on."""
    from oracle.fem_oracle import fem_step

    res = {}
    for follow in (True, False):
        sim, m, P, cons, aim, cms = _c4_scene(1)
        sim.cfg.contact.follow_indenter = follow
        sim.cfg.contact.enable_friction = False
        sim.cfg.newton.velocity_tol = 1e-3
        sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-12
        sim.set_contact_indenters(sim.contact_indenters)
        ind = sim.contact_indenters
        xo, vo = P.copy(), np.zeros_like(P)
        prev = None
        its, pcg = [], []
        for k in range(9):
            if k < 4:
                ind[:, 3] -= 0.35 * float(sim.contact_gaps().amin())  # press ...
            else:
                ind[:, 3] += 1.2e-4                                    # ... then retreat 0.12 mm per step
            cur = ind[0, 1:4].cpu().numpy().copy()
            disp = cur - prev if prev is not None else np.zeros(3)
            prev = cur
            cms[0].ind[1:4] = cur
            sim.step(max_newton_iter=80)
            info = sim.check_step()
            assert len(info["penetrating_envs"]) == 0 and info["newton_iters"].max() < 80
            if k >= 4:
                its.append(int(info["newton_iters"][0])); pcg.append(int(info["pcg_iters"][0]))
            if follow:
                xo, vo, io = fem_step(m, cms[0], xo, vo, cons, aim[0], gravity=sim.cfg.gravity, max_newton=80, velocity_tol=1e-3, pcg_max_iter=600,
                                      pcg_tol_rate=1e-12, coarse=sim.coarse_space, chains=_chains(sim), indenter_disp=disp)
                assert abs(int(info["newton_iters"][0]) - int(io[0])) <= 1, (k, info["newton_iters"], io)
                assert np.abs(sim.x[0].cpu().numpy() - xo).max() <= 2 * 1e-3 * sim.cfg.dt, k
        res[follow] = (sim.x[0].cpu().numpy().copy(), sum(its), sum(pcg))
    print("retreat: Newton / PCG iterations with following", res[True][1:], "without", res[False][1:])
    assert np.abs(res[True][0] - res[False][0]).max() <= 4 * 1e-3 * 0.01  # the same state within the Newton tolerance of both runs
    # (measured 17 / 2 094 against 36 / 3 393; before the edge snap of the Newton iteration the unfollowed loop needed 58 / 4 111)
    assert res[True][1] < res[False][1] and res[True][2] < 0.75 * res[False][2], (res[True][1:], res[False][1:])


def _axle_scene(B, deterministic=False, block_jacobi=False, velocity_tol=2e-3):
    """simple_axle.msh (593 vertices / 2 003 tets, tests/golden/fem_meshes.npz) scaled to 25.8 x 3 x 3 mm, both ends held, a sphere over its middle."""
    from oracle.fem_oracle import ContactModel, FemModel
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg

    g = np.load(Path(__file__).parent / "golden" / "fem_meshes.npz")
    P = (g["simple_axle_points"] - g["simple_axle_points"].min(0)) * 0.01
    T = g["simple_axle_tets"]
    assert len(P) == 593 and len(T) == 2003
    cfg = UipcSimCfg(device="cuda:0")
    if block_jacobi:
        cfg.linear_system.coarse_grid, cfg.linear_system.vertex_chains = None, None
    cfg.linear_system.deterministic = deterministic
    cfg.linear_system.max_iter, cfg.linear_system.tol_rate = 3000, 1e-10
    cfg.newton.velocity_tol = velocity_tol
    cfg.contact.friction_lag = "capped"  # the documented setting for slender bodies (UipcSimCfg.Contact.friction_lag); the gelpad scenes run the default, "ipc"
    sim = UipcSim(cfg, num_envs=B)
    gel = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
    sim.setup_sim(constraint_strength_ratio=1000.0)
    ends = np.where((P[:, 0] < 0.002) | (P[:, 0] > P[:, 0].max() - 0.002))[0]
    aim = np.repeat(P[None], B, 0)
    sim.set_constraints(ends, torch.from_numpy(aim[:, ends]).cuda())
    cons = np.zeros(len(P)); cons[ends] = 1.0
    top, mid = P[:, 2].max(), P[:, 0].max() / 2
    ind = np.zeros((B, 8)); ind[:, 0] = 1.0
    ind[:, 1], ind[:, 2], ind[:, 4] = mid, P[:, 1].max() / 2, 0.004
    ind[:, 3] = top + 0.004 + 0.0009 - 1e-4 * np.arange(B)
    m = FemModel.build(P, T, youngs=gel.cfg.constitution_cfg.youngs_modulus * 1e6, poisson=gel.cfg.constitution_cfg.poisson_rate,
                       density=gel.cfg.mass_density, dt=sim.cfg.dt, strength=1000.0)
    area = gel.surface_vertex_areas()
    kappa = sim.cfg.contact.default_contact_resistance * 1e9 * sim.cfg.contact.d_hat
    cms = [ContactModel(area, ind[b].copy(), sim.cfg.contact.d_hat, kappa, sim.cfg.dt) for b in range(B)]
    return sim, m, P, cons, aim, cms, ind


def test_wide_newton_kernel_steps_simple_axle_with_contact_and_friction():
    """a mesh with MORE vertices than 512 - the reference's simple_axle.msh - stepped by the CU-resident Newton kernel
    (its 768-thread variant: one thread per vertex, state in LDS) with everything the gelpad scene uses: IPC barrier, CCD step bound,
    Coulomb friction (on by default, uipc_sim.py:103-124), the coarse correction on the bounding-box grid of the unstructured mesh and
    the vertex chains found in it.  Against the oracle's fem_step with the same tables: press for three steps, then slide.
    The bent axle has compressed elements with negative curvature: its envs solve some iterations in PSD-safe mode (kFemFlagPsdSafe)."""
    from oracle.fem_oracle import fem_step

    B, vtol = 2, 5e-4
    sim, m, P, cons, aim, cms, ind = _axle_scene(B, velocity_tol=vtol)
    sim.set_contact_indenters(torch.from_numpy(ind))
    indd = sim.contact_indenters
    mu, slide = sim.cfg.contact.default_friction_ratio, 3e-5
    assert sim.cfg.contact.enable_friction and mu > 0
    xo = [P.copy() for _ in range(B)]
    vo = [np.zeros_like(P) for _ in range(B)]
    prev, psd_seen, gap_min = None, False, np.inf
    for k in range(5):
        gap = sim.contact_gaps().amin(1)
        if k < 3:
            indd[:, 3] -= 0.3 * gap
        else:
            indd[:, 1] += slide
            indd[:, 3] += torch.clamp(2 * slide - gap, min=0.0)
        cur = indd[:, 1:4].cpu().numpy().copy()
        disp = cur - prev if prev is not None else np.zeros_like(cur)
        prev = cur
        # (every step starts the oracle from the kernel's state: the bending mode of this rod is nearly free, two runs that both stop
        #  on the Newton tolerance drift apart along it over the steps - 0.1 mm by the fifth - and that drift is not what is compared)
        xo, vo = list(sim.x.cpu().numpy().copy()), list(sim.v.cpu().numpy().copy())
        sim.step(max_newton_iter=60)
        info = sim.check_step()
        assert len(info["penetrating_envs"]) == 0 and len(info["line_search_failed_envs"]) == 0 and int(sim.last_newton_iters) < 60, (k, info)
        psd_seen = psd_seen or len(info["psd_safe_envs"]) > 0
        x = sim.x.cpu().numpy()
        assert np.isfinite(x).all() and float(sim.contact_gaps().amin()) > 0.0
        gap_min = min(gap_min, float(sim.contact_gaps().amin()))
        for b in range(B):
            cms[b].ind[1:4] = cur[b]
            xo[b], vo[b], io = fem_step(m, cms[b], xo[b], vo[b], cons, aim[b], gravity=sim.cfg.gravity, max_newton=60, velocity_tol=vtol,
                                        pcg_max_iter=3000, pcg_tol_rate=1e-10, coarse=sim.coarse_space, chains=_chains(sim),
                                        friction=(mu, sim.cfg.contact.eps_velocity, disp[b]), friction_lag=_lag(sim))
            assert io[0] < 60 and int(io[2]) & 3 == 0, (k, b, io)
            assert np.abs(x[b] - xo[b]).max() <= 2 * vtol * sim.cfg.dt, (k, b, np.abs(x[b] - xo[b]).max(), io, info)  # both inside the Newton tolerance
    assert (P[:, 2] - sim.x[0].cpu().numpy()[:, 2]).max() > 5e-5  # the axle is dented / bent by the sphere
    assert gap_min < sim.cfg.contact.d_hat  # the barrier really acted (the soft rod is pushed away and may swing clear of the zone again)
    assert sim.coarse_space[2].shape[0] == 3 * 16 and len(sim.vertex_chains) > 0  # 3 x 1 x 1 cells over the axle; the chains found in it
    assert psd_seen  # (the safeguard is what this mesh needs: without it the loop crawled into inverted states, oracle and kernel alike)


def test_streaming_newton_kernel_steps_simple_axle_with_sphere_contact():
    """The streaming Newton kernel (state in HBM / L2, any vertex count: what a mesh too large for a CU's LDS, or the deterministic
    switch on one of more than 512, runs on) carries the IPC barrier, the conservative step bound and - since round 5 - Coulomb friction
    with the lag of the step's start (until then the default cfg, friction on, made such a mesh's step fail: "switch friction off");
    chains, the coarse correction, the contact-following start and the edge snap stay with the CU-resident kernel.  simple_axle.msh in
    deterministic mode against the oracle's fem_step with the same block-Jacobi preconditioner, without and with friction (press for
    three steps, then slide): positions, no penetration; and friction really acts (the contact patch is dragged along)."""
    from oracle.fem_oracle import fem_step

    B, slide, sep = 2, 1e-4, 0.0
    for with_friction in (False, True):
        sim, m, P, cons, aim, cms, ind = _axle_scene(B, deterministic=True, block_jacobi=True, velocity_tol=5e-4)
        sim.cfg.contact.enable_friction = with_friction
        sim.set_contact_indenters(torch.from_numpy(ind))
        indd = sim.contact_indenters
        mu, vtol = sim.cfg.contact.default_friction_ratio, sim.cfg.newton.velocity_tol
        xo = [P.copy() for _ in range(B)]
        vo = [np.zeros_like(P) for _ in range(B)]
        prev = None
        for k in range(5):
            gap = sim.contact_gaps().amin(1)
            if k < 3:
                indd[:, 3] -= 0.3 * gap
            else:
                indd[:, 1] += slide
                indd[:, 3] += torch.clamp(2 * slide - gap, min=0.0)
            cur = indd[:, 1:4].cpu().numpy().copy()
            disp = cur - prev if prev is not None else np.zeros_like(cur)
            prev = cur
            # (as in the wide-kernel test: every step starts the oracle from the kernel's state - the rod's bending mode is nearly free)
            xo, vo = list(sim.x.cpu().numpy().copy()), list(sim.v.cpu().numpy().copy())
            sim.step(max_newton_iter=30)
            assert sim.newton_kernel_resident is False  # the streaming kernel
            info = sim.check_step()
            assert len(info["penetrating_envs"]) == 0 and len(info["line_search_failed_envs"]) == 0, (with_friction, k, info)
            x = sim.x.cpu().numpy()
            assert np.isfinite(x).all() and float(sim.contact_gaps().amin()) > 0.0
            for b in range(B):
                cms[b].ind[1:4] = cur[b]
                if with_friction and k >= 3 and b == 0:  # what the same step looks like WITHOUT friction: the comparison below must be able to tell
                    x_nf, _, _ = fem_step(m, cms[b], xo[b], vo[b], cons, aim[b], gravity=sim.cfg.gravity, max_newton=30, velocity_tol=vtol,
                                          pcg_max_iter=3000, pcg_tol_rate=1e-10, coarse=None, chains=None, lag_prec=False, indenter_disp=np.zeros(3))
                    sep = max(sep, float(np.abs(x[b] - x_nf).max()))
                xo[b], vo[b], io = fem_step(m, cms[b], xo[b], vo[b], cons, aim[b], gravity=sim.cfg.gravity, max_newton=30, velocity_tol=vtol,
                                            pcg_max_iter=3000, pcg_tol_rate=1e-10, coarse=None, chains=None, lag_prec=False,
                                            friction=(mu, sim.cfg.contact.eps_velocity, disp[b]) if with_friction else None,
                                            indenter_disp=np.zeros(3), friction_lag=_lag(sim))  # (no contact-following start in the streaming kernel)
                assert io[0] < 30 and int(io[2]) & 3 == 0, (with_friction, k, b, io)
                assert np.abs(x[b] - xo[b]).max() <= 2 * vtol * sim.cfg.dt, (with_friction, k, b, np.abs(x[b] - xo[b]).max(), io)  # both inside the Newton tolerance
        assert (P[:, 2] - sim.x[0].cpu().numpy()[:, 2]).max() > 5e-5  # the axle is dented / bent by the sphere
    # friction really acted: the frictional kernel state lies further from the oracle's FRICTIONLESS solve of the same step than the tolerance
    # it matches the frictional one to
    assert sep > 3 * 2 * 5e-4 * 0.01, sep


def test_deterministic_and_atomic_sweeps_agree_and_deterministic_runs_are_bit_identical():
    """`UipcSimCfg.linear_system.deterministic` (tacex_fem_set_deterministic): the window + CSR-gather sweeps give bit-identical runs;
    the default LDS-atomic sweeps (ds_add_f64, timing-dependent summation order) land on the same state to round-off - same Newton and
    PCG iteration counts, positions within 1e-9 of the mesh size."""
    outs = {}
    for det in (True, True, False):
        sim, m, P, cons, aim, cms = _c4_scene(2)
        sim.cfg.linear_system.deterministic = det
        _lib_check = sim._lib.tacex_fem_set_deterministic(sim._handle, 1 if det else 0)
        assert _lib_check == 0
        ind = sim.contact_indenters
        for k in range(4):
            ind[:, 3] -= 0.3 * sim.contact_gaps().amin(1)
            ind[:, 1] += 2e-5
            sim.step(max_newton_iter=40)
        outs.setdefault(det, []).append((sim.x.cpu().numpy().copy(), sim.step_info.cpu().numpy().copy()))
    (x0, i0), (x1, i1) = outs[True]
    np.testing.assert_array_equal(x0, x1)
    np.testing.assert_array_equal(i0, i1)
    xa, ia = outs[False][0]
    assert np.abs(xa - x0).max() <= 1e-9 * np.ptp(P) + 2 * 0.05 * 0.01 * (ia[:, 0] != i0[:, 0]).any()
    assert np.abs(ia[:, 0] - i0[:, 0]).max() <= 1 and np.abs(ia[:, 3] - i0[:, 3]).max() <= 3
