"""GPU parity tests of the reference's own UIPC scene (second slice: csrc/fem_ball.h through `UipcSim` with an
`AffineBodyConstitutionCfg` object) against oracle/abd_oracle.py - PARITY UNPINNED like every FEM row (libuipc is not in the reference
tree); the oracle pins itself in tests/test_abd_oracle.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(B=2, mesh=(6, 8, 2), R=0.006, level=1, press=4e-4, ground_gap=0.6, density=1e3, shift=(0.0005, 0.0005), dhat=5e-4, gh=0.001,
           velocity_tol=None, tol_rate=None, transrate_tol=None):
    """The same scene twice: `UipcSim` (HIP) and `BallScene` (oracle).  press: how far the pad's face sits inside the ball's barrier zone
    (negative: outside); ground_gap: the ball's lowest point above the ground in units of d_hat."""
    from oracle.abd_oracle import AffineBody, BallScene
    from oracle.fem_oracle import FemModel
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.gelpad_scene import icosphere
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    P, T = gelpad_box_mesh(*mesh)
    size = P.max(0) - P.min(0)
    Pw = P * np.array([1.0, -1.0, -1.0]) + np.array([-size[0] / 2 + shift[0], size[1] / 2 + shift[1], 0.0])
    zc = gh + dhat * ground_gap + R
    Pw[:, 2] += zc + R + (dhat - press) - Pw[:, 2].min()
    cfg = UipcSimCfg(device="cuda:0")
    cfg.contact.d_hat, cfg.ground_height = dhat, gh
    if velocity_tol is not None:
        cfg.newton.velocity_tol = velocity_tol
    if transrate_tol is not None:
        cfg.newton.transrate_tol = transrate_tol
    if tol_rate is not None:
        cfg.linear_system.tol_rate = tol_rate
        cfg.linear_system.max_iter = 4000
    sim = UipcSim(cfg, num_envs=B)
    pad = UipcObject(UipcObjectCfg(mesh_points=Pw, mesh_tets=T), sim)
    vb, tb = icosphere(R, level)
    UipcObject(UipcObjectCfg(mesh_points=vb, mesh_tris=tb, mass_density=density, init_pos=(0.0, 0.0, zc),
                             constitution_cfg=UipcObjectCfg.AffineBodyConstitutionCfg()), sim)
    sim.setup_sim(constraint_strength_ratio=1000.0)
    back = np.where(Pw[:, 2] > Pw[:, 2].max() - 1e-12)[0]
    sim.set_constraints(back, torch.from_numpy(np.repeat(Pw[None, back], B, 0)).cuda())
    m = FemModel.build(Pw, T, youngs=pad.cfg.constitution_cfg.youngs_modulus * 1e6, poisson=pad.cfg.constitution_cfg.poisson_rate,
                       density=pad.cfg.mass_density, dt=cfg.dt, strength=1000.0)
    sc = BallScene(m, pad.surface_triangles(), pad.surface_vertex_areas(), AffineBody(vb, tb, density=density), dhat=dhat, ground_height=gh,
                   resistance=cfg.contact.default_contact_resistance)
    if cfg.contact.enable_friction:  # one default contact model for every pair of surfaces (US:192-201)
        sc.mu, sc.eps_v = cfg.contact.default_friction_ratio, cfg.contact.eps_velocity
    cons = np.zeros(len(Pw))
    cons[back] = 1.0
    return sim, sc, cons, back


def _y(sim, b):
    return np.concatenate([sim.x[b].cpu().numpy(), sim.q[b].cpu().numpy()], 0)


def test_moments_of_the_ball_mesh_vs_oracle():
    import ctypes as C

    sim, sc, cons, back = _build()
    S = (C.c_double * 16)()
    kv = C.c_double()
    from tacex_amd import _lib
    _lib.check(sim._lib.tacex_fem_ball_moments(sim._handle, S, C.byref(kv)), "moments")
    assert np.allclose(np.array(S).reshape(4, 4), sc.ball.S, rtol=1e-12, atol=1e-22)
    assert kv.value == pytest.approx(sc.ball.kv, rel=1e-12)


def test_energy_and_gradient_of_the_step_potential_vs_oracle():
    """Every term at once, in a state with pad-vertex / ball-triangle pairs, ball-vertex / pad-triangle pairs, the ground under the ball,
    a stretched and sheared body (orthogonality energy), velocities (inertia), constraint offsets and lagged friction of every contact
    (sliding relative to a previous state 30 um / 300 um away: stick and slip): energy to 1e-10 relative, the gradient of all V + 4 rows
    to 1e-9 of its largest entry; env 1 is a second, different state."""
    sim, sc, cons, back = _build()
    V = sc.V
    rng = np.random.default_rng(5)
    y = [_y(sim, b) + 2e-5 * rng.standard_normal((V + 4, 3)) for b in range(2)]
    y[1][V + 1:] += 1e-3 * rng.standard_normal((3, 3))
    yt = [yy + 1e-5 * rng.standard_normal(yy.shape) for yy in y]
    aim = sim.aim_position.cpu().numpy() + 1e-5
    sim.aim_position.copy_(torch.from_numpy(aim).cuda())
    k0 = sc.pairs(y[0])
    assert len(k0[0][0]) >= 1 and len(k0[1][0]) >= 1 and len(k0[2][0]) >= 1  # both point-triangle kinds and edge-edge pairs
    x = torch.from_numpy(np.stack([yy[:V] for yy in y])).cuda()
    q = torch.from_numpy(np.stack([yy[V:] for yy in y])).cuda()
    xt = torch.from_numpy(np.stack([yy[:V] for yy in yt])).cuda()
    qt = torch.from_numpy(np.stack([yy[V:] for yy in yt])).cuda()
    yp = [y[0] + 3e-5 * rng.standard_normal(y[0].shape), y[1] + 3e-4 * rng.standard_normal(y[1].shape)]
    for b in range(2):
        yp[b][V + 1:] = y[b][V + 1:] + (1e-3 if b == 0 else 2e-2) * rng.standard_normal((3, 3))
    xp = torch.from_numpy(np.stack([yy[:V] for yy in yp])).cuda()
    qp = torch.from_numpy(np.stack([yy[V:] for yy in yp])).cuda()
    E, g, si = sim.ball_terms(x, q, xt, qt, x_prev=xp, q_prev=qp)
    assert int(si[:, 2].max()) == 0 and sc.mu == 0.5
    slid = []
    for b in range(2):
        sc._lag = sc.friction_lag(y[b])[:4] + (yp[b],)  # forces / normals / weights of the state itself, sliding measured from yp
        slid.append(np.linalg.norm(sc._fric(y[b])[3], axis=1))
        Eo = sc.energy(y[b], yt[b], cons, aim[b])
        go = sc.gradient(y[b], yt[b], cons, aim[b])
        assert abs(float(E[b]) - Eo) <= 1e-10 * abs(Eo), (b, float(E[b]), Eo)
        gk = g[b].cpu().numpy()
        assert np.abs(gk - go).max() <= 1e-9 * np.abs(go).max(), (b, np.abs(gk - go).max(), np.abs(go).max())
    eps = sc.eps_v * sc.dt
    assert (slid[0] < eps).any() and (slid[1] > eps).any()  # both branches of the friction potential were in play
    sc._lag = None


def test_edge_edge_pairs_mollified_and_switched_off_vs_oracle():
    """Edge-edge pairs on their own: (i) env 0 holds a pad edge laid nearly parallel over a ball edge - the pair is mollified (m < 1) and the
    mollifier's own gradient is in play; (ii) with `tacex_fem_set_edge_edge(ctx, 0)` the kernel's terms are the oracle's without that pair
    kind, and differ from (i)."""
    from tacex_amd import _lib

    sim, sc, cons, back = _build(press=3.5e-4)
    V = sc.V
    y = [_y(sim, b) for b in range(2)]
    ee = sc.pairs(y[0])[2]
    assert len(ee[0]) > 0
    k = int(np.argmin(ee[3]))
    pe, be = sc.pad_edges[ee[0][k]], sc.ball_edges[ee[1][k]]
    xb = sc.ball.points(y[0][V:])
    e2 = xb[be[1]] - xb[be[0]]
    u2 = e2 / np.linalg.norm(e2)
    nn = ee[4][k] - (ee[4][k] @ u2) * u2
    nn /= np.linalg.norm(nn)
    mid = xb[be[0]] + 0.5 * e2 + 0.4 * sc.dhat * nn
    L = np.linalg.norm(y[0][pe[1]] - y[0][pe[0]])
    skew = 0.01 * L * np.cross(u2, nn)
    y[0][pe[0]] = mid - 0.3 * L * u2 - skew
    y[0][pe[1]] = mid + 0.3 * L * u2 + skew
    mol = sc._pair_rows(y[0])[5]
    assert (mol < 1.0).any() and sc._pair_rows(y[0])[6] is not None
    yt = [yy + 1e-5 for yy in y]
    aim = sim.aim_position.cpu().numpy()
    x = torch.from_numpy(np.stack([yy[:V] for yy in y])).cuda()
    q = torch.from_numpy(np.stack([yy[V:] for yy in y])).cuda()
    xt = torch.from_numpy(np.stack([yy[:V] for yy in yt])).cuda()
    qt = torch.from_numpy(np.stack([yy[V:] for yy in yt])).cuda()
    mu = sc.mu
    sc.mu = 0.0  # (no x_prev / q_prev handed over: the kernel's terms are frictionless)
    try:
        E, g, si = sim.ball_terms(x, q, xt, qt)
        assert int(si[:, 2].max()) == 0
        on = []
        for b in range(2):
            Eo, go = sc.energy(y[b], yt[b], cons, aim[b]), sc.gradient(y[b], yt[b], cons, aim[b])
            on.append(Eo)
            assert abs(float(E[b]) - Eo) <= 1e-10 * abs(Eo), (b, float(E[b]), Eo)
            assert np.abs(g[b].cpu().numpy() - go).max() <= 1e-9 * np.abs(go).max(), b
        _lib.check(sim._lib.tacex_fem_set_edge_edge(sim._handle, 0), "tacex_fem_set_edge_edge")
        sc.edge_edge = False
        E, g, si = sim.ball_terms(x, q, xt, qt)
        for b in range(2):
            Eo, go = sc.energy(y[b], yt[b], cons, aim[b]), sc.gradient(y[b], yt[b], cons, aim[b])
            assert Eo < on[b] * (1 - 1e-6)  # the edge-edge barrier energy is gone
            assert abs(float(E[b]) - Eo) <= 1e-10 * abs(Eo), (b, float(E[b]), Eo)
            assert np.abs(g[b].cpu().numpy() - go).max() <= 1e-9 * np.abs(go).max(), b
    finally:
        sc.mu, sc.edge_edge = mu, True


def test_step_vs_oracle_step_from_outside_every_barrier_zone():
    """Three backward-Euler steps of the back face moving down onto the ball (denser ball: tests/test_abd_oracle.py says why), both solvers
    run to a tight tolerance: same end states (pad vertices and the ball's twelve unknowns) step by step, no ground or pair gap closed."""
    sim, sc, cons, back = _build(B=2, press=-2e-5, ground_gap=1.02, density=1e5, shift=(0.0008, 0.0005), velocity_tol=1e-6, transrate_tol=1e-5,
                                 tol_rate=1e-12)
    V = sc.V
    yo = [_y(sim, b) for b in range(2)]
    vo = [np.zeros_like(yo[0]) for _ in range(2)]
    aim0 = sim.aim_position.clone()
    depth = np.array([5e-5, 8e-5])
    for k in range(3):
        aim = aim0.clone()
        aim[:, :, 2] -= torch.from_numpy(depth * (k + 1)).cuda()[:, None]
        sim.aim_position.copy_(aim)
        sim.step(max_newton_iter=60)
        info = sim.check_step()
        assert len(info["penetrating_envs"]) == 0 and len(info["line_search_failed_envs"]) == 0 and len(info["pair_list_overflow_envs"]) == 0, info
        assert info["newton_iters"].max() < 60
        for b in range(2):
            yo[b], vo[b], io = sc.step(yo[b], vo[b], cons, aim[b].cpu().numpy(), gravity=sim.cfg.gravity, max_newton=60, velocity_tol=1e-6,
                                       transrate_tol=1e-5, pcg_max_iter=4000, pcg_tol_rate=1e-12)
            assert io[0] < 60 and int(io[2]) == 0
            yk = _y(sim, b)
            assert np.abs(yk[:V + 1] - yo[b][:V + 1]).max() <= 2e-8, (k, b, np.abs(yk[:V + 1] - yo[b][:V + 1]).max())  # 1e-6 m/s * dt = 1e-8 m each
            assert np.abs(yk[V + 1:] - yo[b][V + 1:]).max() <= 2e-7, (k, b)
            assert (sc.ball.points(yk[V:])[:, 2] > sc.gh).all()
    assert yo[0][V, 2] < sim.cfg.ground_height + 1.02 * 5e-4 + 0.006  # the ball was pushed into the ground's barrier zone
    kinds = sc.pairs(_y(sim, 1))
    assert len(kinds[0][0]) + len(kinds[1][0]) >= 1


def test_reference_scene_at_default_tolerances_runs_clean_and_ends_stationary_when_solved_tightly():
    """`FemBallScene` (what bench.py's c4_ball entry steps): the C4 pad over the reference's ball on the ground, a press-and-release period.
    At the reference's default tolerances and contact model (uipc_sim.py:57-124: friction ratio 0.5) every env converges below the iteration
    cap with no flag (ground, line search, list overflow); the ball ends lower while pressed and the pad's face never crosses it.  And the end state of a tightly solved step is a stationary point of the
    plain incremental potential (oracle gradient, no solver code shared): below 1e-5 of the largest pair force."""
    from oracle.abd_oracle import AffineBody, BallScene
    from oracle.fem_oracle import FemModel, barrier
    from tacex_amd.uipc.gelpad_scene import FemBallScene
    from tacex_amd.uipc.uipc_sim import UipcSimCfg

    B = 4
    sc_d = FemBallScene(B, "cuda:0", max_newton_iter=64)
    z0 = sc_d.sim.q[:, 0, 2].clone()
    zmin = z0.clone()
    worst_iters, pressed_max = 0, 0.0
    P0 = sc_d.gelpad.points
    low = np.where(P0[:, 2] < P0[:, 2].min() + 1e-12)[0]
    face = int(low[np.argmin(np.hypot(P0[low, 0], P0[low, 1]))])  # the contact-face vertex nearest the ball's axis
    for i in range(12):
        sc_d.step(i)
        info = sc_d.sim.check_step()
        assert len(info["penetrating_envs"]) == 0 and len(info["pair_list_overflow_envs"]) == 0, (i, info)
        assert len(info["line_search_failed_envs"]) == 0 and info["newton_iters"].max() < 64, (i, info)
        worst_iters = max(worst_iters, int(info["newton_iters"].max()))
        zmin = torch.minimum(zmin, sc_d.sim.q[:, 0, 2])
        assert torch.isfinite(sc_d.sim.x).all() and torch.isfinite(sc_d.sim.q).all()
        thick = sc_d.sim.x[:, :, 2].amax(1) - sc_d.sim.x[:, face, 2]  # back face to the contact-face vertex over the ball
        pressed_max = max(pressed_max, float((0.0045 - thick).max()))
    print(f"default tolerances: worst Newton iteration count of any env and step {worst_iters} (cap 64)")
    # the ball rests on the ground's barrier (a 10 GPa wall for its 30 mN): pressed, it sinks by fractions of a micron and the soft pad takes the rest
    assert float((z0 - zmin).min()) > 1e-7 and float((z0 - zmin).max()) < 5e-5
    assert pressed_max > 2e-4  # the deepest env's pad was squeezed by more than 0.2 mm under the ball
    # tight solve of one more step, checked against the oracle's plain gradient
    cfg = UipcSimCfg(device="cuda:0")
    cfg.newton.velocity_tol, cfg.newton.transrate_tol, cfg.linear_system.tol_rate, cfg.linear_system.max_iter = 1e-7, 1e-6, 1e-12, 4000
    t = FemBallScene(2, "cuda:0", max_newton_iter=200, cfg=cfg, ball_density=1e5)
    sim, pad, ball = t.sim, t.gelpad, t.ball
    m = FemModel.build(pad.points, pad.tets, youngs=pad.cfg.constitution_cfg.youngs_modulus * 1e6, poisson=pad.cfg.constitution_cfg.poisson_rate,
                       density=pad.cfg.mass_density, dt=cfg.dt, strength=1000.0)
    osc = BallScene(m, pad.surface_triangles(), pad.surface_vertex_areas(), AffineBody(ball.points, ball.tris, density=1e5), dhat=cfg.contact.d_hat,
                    ground_height=cfg.ground_height, resistance=cfg.contact.default_contact_resistance)
    osc.mu, osc.eps_v = cfg.contact.default_friction_ratio, cfg.contact.eps_velocity
    worst, checked = 0.0, 0
    for i in range(10):
        y_n = [np.concatenate([sim.x[b].cpu().numpy(), sim.q[b].cpu().numpy()]) for b in range(2)]
        v_n = [np.concatenate([sim.v[b].cpu().numpy(), sim.qv[b].cpu().numpy()]) for b in range(2)]
        t.step(i)
        info = sim.check_step()
        assert info["newton_iters"].max() < 200 and int(sim.step_info[:, 2].max()) == 0, (i, info)
        cons = sim.is_constrained[0].cpu().numpy().astype(np.float64)
        for b in range(2):
            yt = y_n[b] + cfg.dt * v_n[b]
            g3 = cfg.dt**2 * np.asarray(cfg.gravity)
            yt[:osc.V] += g3
            yt[osc.V] += g3
            y = np.concatenate([sim.x[b].cpu().numpy(), sim.q[b].cpu().numpy()])
            osc._lag = osc.friction_lag(y_n[b])  # IPC's lag: the contacts of the state the step started from
            g = osc.gradient(y, yt, cons, sim.aim_position[b].cpu().numpy())
            _, _, w, d, _, mol, _ = osc._pair_rows(y)  # point-triangle pairs of both kinds and edge-edge pairs
            w = w * mol
            scale = cfg.dt**2 * osc.kappa * np.abs(w * barrier(d / osc.dhat)[1] / osc.dhat).max() if len(d) else 0.0
            if scale < 1e-6:  # (a pair that has only just entered the zone pushes with less than the round-off of the pad's elastic forces)
                assert np.abs(g).max() <= 1e-11, (i, b, np.abs(g).max())
                continue
            checked += 1
            worst = max(worst, np.abs(g).max() / scale)
            assert np.abs(g).max() <= 1e-5 * scale, (i, b, np.abs(g).max(), scale)
    assert checked >= 2  # states with real pair forces were among them
    print(f"worst |grad| / pair force over the tight run: {worst:.2e}")


def test_reset_of_single_envs_puts_pad_and_ball_back():
    """`UipcSim.reset(env_ids)` in a scene with an affine body (uipc_object.py:280-370): the listed envs' pads AND balls return to where the scene
    placed them, at rest; the others keep their state; the next steps run clean."""
    from tacex_amd.uipc.gelpad_scene import FemBallScene

    B = 4
    sc = FemBallScene(B, "cuda:0", max_newton_iter=64)
    for i in range(8):
        sc.step(i)
    rest = torch.from_numpy(sc.gelpad.points).cuda()
    q0 = sc.sim._q0
    xb, qb = sc.sim.x.clone(), sc.sim.q.clone()
    assert float((xb[3] - rest).abs().max()) > 1e-5 and float((qb[3] - q0).abs().max()) > 1e-8
    sc.sim.reset([1, 3])
    assert torch.equal(sc.sim.x[[1, 3]], rest[None].expand(2, -1, -1)) and torch.equal(sc.sim.q[[1, 3]], q0[None].expand(2, -1, -1))
    assert float(sc.sim.v[[1, 3]].abs().max()) == 0.0 and float(sc.sim.qv[[1, 3]].abs().max()) == 0.0
    assert torch.equal(sc.sim.x[[0, 2]], xb[[0, 2]]) and torch.equal(sc.sim.q[[0, 2]], qb[[0, 2]])
    for i in range(8, 11):
        sc.step(i)
        info = sc.sim.check_step()
        assert len(info["line_search_failed_envs"]) == 0 and len(info["pair_list_overflow_envs"]) == 0 and info["newton_iters"].max() < 64, (i, info)


def test_kinematic_body_is_fixed_within_a_step_and_dents_the_pad():
    """`AffineBodyConstitutionCfg(kinematic=True)` (uipc_object.py:70-73, 463-466): the ball's twelve unknowns do not move in a step whatever
    pushes it - no gravity fall, no yielding to the pad - and a ball the caller lifts into the pad dents it (pairs both ways act on the pad)."""
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.gelpad_scene import icosphere
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    B, R, dhat, gh = 2, 0.006, 5e-4, 0.001
    P, T = gelpad_box_mesh(6, 8, 2)
    size = P.max(0) - P.min(0)
    Pw = P * np.array([1.0, -1.0, -1.0]) + np.array([-size[0] / 2 + 0.0008, size[1] / 2 + 0.0005, 0.0])
    zc = gh + 0.002 + R
    Pw[:, 2] += zc + R + 1.02 * dhat - Pw[:, 2].min()
    cfg = UipcSimCfg(device="cuda:0")
    cfg.contact.d_hat, cfg.ground_height = dhat, gh
    sim = UipcSim(cfg, num_envs=B)
    pad = UipcObject(UipcObjectCfg(mesh_points=Pw, mesh_tets=T), sim)
    vb, tb = icosphere(R, 1)
    UipcObject(UipcObjectCfg(mesh_points=vb, mesh_tris=tb, init_pos=(0.0, 0.0, zc),
                             constitution_cfg=UipcObjectCfg.AffineBodyConstitutionCfg(kinematic=True)), sim)
    sim.setup_sim(constraint_strength_ratio=1000.0)
    back = np.where(Pw[:, 2] > Pw[:, 2].max() - 1e-12)[0]
    sim.set_constraints(back, torch.from_numpy(np.repeat(Pw[None, back], B, 0)).cuda())
    q0 = sim.q.clone()
    face = int(np.argmin(np.hypot(Pw[:, 0], Pw[:, 1]) + 1e3 * (Pw[:, 2] > Pw[:, 2].min() + 1e-12)))  # contact-face vertex nearest the ball's axis
    z_face0 = float(sim.x[0, face, 2])
    for k in range(8):
        sim.q[:, 0, 2] += 1e-4 * torch.tensor([1.0, 0.5], device="cuda", dtype=torch.float64)  # the caller lifts the ball: 0.8 / 0.4 mm in all
        q_before = sim.q.clone()
        sim.step(max_newton_iter=64)
        info = sim.check_step()
        assert len(info["line_search_failed_envs"]) == 0 and len(info["pair_list_overflow_envs"]) == 0 and info["newton_iters"].max() < 64, (k, info)
        assert torch.equal(sim.q, q_before)  # fixed within the step: no fall, no yielding
    lift = sim.x[:, face, 2].cpu().numpy() - z_face0
    assert lift[0] > 1.5e-4 and lift[0] > lift[1] > 0.0, lift  # the pad's face was pushed up, more where the ball rose further
    gaps = sim.x[:, face, 2].cpu().numpy() - (sim.q[:, 0, 2].cpu().numpy() + R)
    assert (gaps > 0).all()


def test_scene_guards_of_the_affine_body_path():
    """What the ball path does not do is refused loudly: prescribed indenters next to a body, the deterministic switch, a body without contact,
    a second body; a body given by tets takes their surface."""
    from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
    from tacex_amd.uipc.gelpad_scene import FemBallScene, icosphere
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    sc = FemBallScene(2, "cuda:0")
    with pytest.raises(NotImplementedError):
        sc.sim.set_contact_indenters(torch.zeros((2, 8), dtype=torch.float64, device="cuda"))
    P, T = gelpad_box_mesh(4, 5, 2)
    vb, tb = icosphere(0.005, 1)
    for bad in ("deterministic", "no_contact", "two_bodies"):
        cfg = UipcSimCfg(device="cuda:0")
        if bad == "deterministic":
            cfg.linear_system.deterministic = True
        if bad == "no_contact":
            cfg.contact.enable = False
        sim = UipcSim(cfg, num_envs=1)
        UipcObject(UipcObjectCfg(mesh_points=P + np.array([0, 0, 0.02]), mesh_tets=T), sim)
        UipcObject(UipcObjectCfg(mesh_points=vb, mesh_tris=tb, init_pos=(0.01, 0.012, 0.007), constitution_cfg=UipcObjectCfg.AffineBodyConstitutionCfg()), sim)
        if bad == "two_bodies":
            UipcObject(UipcObjectCfg(mesh_points=vb, mesh_tris=tb, constitution_cfg=UipcObjectCfg.AffineBodyConstitutionCfg()), sim)
        with pytest.raises((NotImplementedError, RuntimeError)):
            sim.setup_sim()
    # a body given as a tet mesh: its boundary faces are the surface
    Pb, Tb = gelpad_box_mesh(2, 2, 2, size=(0.006, 0.006, 0.006))
    sim = UipcSim(UipcSimCfg(device="cuda:0", ground_height=0.001), num_envs=1)
    UipcObject(UipcObjectCfg(mesh_points=P + np.array([0, 0, 0.02]), mesh_tets=T), sim)
    body = UipcObject(UipcObjectCfg(mesh_points=Pb - 0.003, mesh_tets=Tb, init_pos=(0.01, 0.012, 0.0045),
                                    constitution_cfg=UipcObjectCfg.AffineBodyConstitutionCfg()), sim)
    assert body.tris.shape == (48, 3)
    sim.setup_sim()
    import ctypes as C
    S = (C.c_double * 16)()
    sim._lib.tacex_fem_ball_moments(sim._handle, S, None)
    assert np.array(S)[0] == pytest.approx(1e3 * 0.006**3, rel=1e-9)  # mass of the cube
    sim.step(max_newton_iter=30)  # falls into the ground's barrier zone and is held
    assert len(sim.check_step()["line_search_failed_envs"]) == 0 and float(sim.q[0, 0, 2]) > 0.001 + 0.003


def test_candidate_list_overflow_is_flagged_not_fatal():
    """A scene outside what the fixed-size lists of csrc/fem_ball.h hold (a barrier zone of 6 mm around a level-3 ball: thousands of pairs
    inside the reach of the lists, capacities 512 candidates / 4096 pairs / 1024 active): the lists clamp, `check_step()` names the envs
    (flag 16), nothing is written out of bounds - the neighbouring env's state and the workspace behind it are untouched, positions stay finite."""
    from tacex_amd.uipc.gelpad_scene import FemBallScene

    sc = FemBallScene(3, "cuda:0", max_newton_iter=3, level=3, d_hat=6e-3)
    guard = torch.full((4096,), 7.25, dtype=torch.float64, device="cuda:0")  # (allocated right behind: a stray write would likely land here or fault)
    for i in range(2):
        sc.step(i)
    torch.cuda.synchronize()
    info = sc.sim.check_step(raise_on_penetration=False)
    assert len(info["pair_list_overflow_envs"]) == 3, info
    assert torch.isfinite(sc.sim.x).all() and torch.isfinite(sc.sim.q).all()
    assert bool((guard == 7.25).all())
