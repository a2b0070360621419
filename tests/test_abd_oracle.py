"""Known-answer tests of oracle/abd_oracle.py (CPU; the oracle of SURVEY 8f n4's first slice is PARITY UNPINNED - libuipc is absent - so it
pins itself): mass moments of the sphere, point-triangle distance derivatives in every region, finite-difference consistency of every
term of the scene's incremental potential, symmetry / positive semi-definiteness of the Gauss-Newton operator and its diagonal blocks,
free fall, rest on the ground (barrier force = weight), action = reaction between pad and ball, and a pressed pad converging to a
stationary point without penetration."""
import numpy as np
import pytest

from oracle.abd_oracle import AffineBody, BallScene, icosphere, point_triangle
from oracle.fem_oracle import FemModel, box_tet_mesh


def _surface(P, T):
    faces = {}
    for t in T:
        for f in ((t[0], t[2], t[1]), (t[0], t[1], t[3]), (t[1], t[2], t[3]), (t[0], t[3], t[2])):
            key = tuple(sorted(f))
            faces[key] = None if key in faces else f
    tri = np.asarray([f for f in faces.values() if f is not None], np.int64)
    a = 0.5 * np.linalg.norm(np.cross(P[tri[:, 1]] - P[tri[:, 0]], P[tri[:, 2]] - P[tri[:, 0]]), axis=1)
    w = np.zeros(len(P))
    np.add.at(w, tri.reshape(-1), np.repeat(a / 3.0, 3))
    return tri, w


def _scene(press=4e-4, level=1, R=0.006, dhat=5e-4, mesh=(6, 8, 2), gh=0.001, shift=(0.0005, 0.0005), pole=False, ground_gap=0.6):
    """Pad turned contact-face-down over a ball that rests on the ground; `press`: how far the pad's face sits inside the ball's barrier zone."""
    P, T = box_tet_mesh(*mesh)
    m = FemModel.build(P, T, dt=0.01, strength=1000.0)
    P, T = m.X, m.tets
    Pw = P * np.array([1.0, -1.0, -1.0])  # rotated by pi about x: the contact face (z = max) looks down
    Pw += np.array([-P[:, 0].max() / 2 + shift[0], P[:, 1].max() / 2 + shift[1], 0.0])  # (off the symmetric spot: a pad vertex exactly over a ball
                                                                                          #  vertex sits on the region boundaries of six triangles)
    zc = gh + dhat * ground_gap + R  # ball centre: its lowest point `ground_gap` d_hat above the ground
    Pw[:, 2] += zc + R + (dhat - press) - Pw[:, 2].min()
    m.X = Pw  # (rest shape = world placement: a rotation leaves dm_inv's products F unchanged only if rebuilt)
    m = FemModel.build(Pw, T, dt=0.01, strength=1000.0)
    tri, area = _surface(m.X, m.tets)
    vb, tb = icosphere(R, level)
    if pole:  # turn the mesh so that a vertex (and, the mesh being point-symmetric, its antipode) lies on the z axis: contact normals through the centre
        a = vb[0] / np.linalg.norm(vb[0])
        vx = np.cross(a, [0.0, 0.0, 1.0])
        c = a[2]
        K = np.array([[0, -vx[2], vx[1]], [vx[2], 0, -vx[0]], [-vx[1], vx[0], 0]])
        vb = vb @ (np.eye(3) + K + K @ K / (1.0 + c)).T
    ball = AffineBody(vb, tb)
    sc = BallScene(m, tri, area, ball, dhat=dhat, ground_height=gh)
    y = np.concatenate([m.X, AffineBody.rest_q([0.0, 0.0, zc])], 0)
    cons = (m.X[:, 2] > m.X[:, 2].max() - 1e-9).astype(np.float64)  # the back face (now on top) is held
    return sc, y, cons


def test_sphere_moments_and_areas():
    R = 0.009
    v, t = icosphere(R, 3)
    b = AffineBody(v, t, density=1e3)
    assert abs(b.vol / (4 / 3 * np.pi * R**3) - 1) < 0.01
    assert np.abs(b.S[0, 1:]).max() < 1e-12 * b.S[0, 0]
    iso = 1e3 * 4 * np.pi * R**5 / 15
    assert np.abs(b.S[1:, 1:] - iso * np.eye(3)).max() < 0.02 * iso
    assert abs(b.area.sum() / (4 * np.pi * R**2) - 1) < 0.01
    assert b.kv == pytest.approx(100e6 * b.vol)


def test_point_triangle_regions_and_derivatives():
    rng = np.random.default_rng(0)
    a, b, c = np.array([[0.0, 0, 0]]), np.array([[1.0, 0, 0]]), np.array([[0.2, 0.9, 0]])
    pts = rng.uniform(-1.5, 2.0, (400, 3))
    beta, d, n = point_triangle(pts, a, b, c)
    assert np.allclose(beta.sum(-1), 1.0) and (beta > -1e-12).all()
    kinds = {tuple((bb > 1e-9).astype(int)) for bb in beta[:, 0]}
    assert len(kinds) == 7  # three vertices, three edges, the face
    h = 1e-7
    for k in range(0, 400, 7):
        for (arr, sign_beta) in ((None, None), (a, 0), (b, 1), (c, 2)):
            for i in range(3):
                e = np.zeros(3); e[i] = h
                if arr is None:
                    dp = point_triangle(pts[k:k + 1] + e, a, b, c)[1][0, 0] - point_triangle(pts[k:k + 1] - e, a, b, c)[1][0, 0]
                    want = n[k, 0, i]
                else:
                    tri = [a.copy(), b.copy(), c.copy()]
                    tri[sign_beta] = tri[sign_beta] + e
                    dpl = point_triangle(pts[k:k + 1], *tri)[1][0, 0]
                    tri[sign_beta] = tri[sign_beta] - 2 * e
                    dp = dpl - point_triangle(pts[k:k + 1], *tri)[1][0, 0]
                    want = -beta[k, 0, sign_beta] * n[k, 0, i]
                assert abs(dp / (2 * h) - want) < 1e-5, (k, sign_beta, i)


def test_orthogonality_energy_gradient_and_gauss_newton_hessian():
    v, t = icosphere(0.006, 1)
    b = AffineBody(v, t)
    rng = np.random.default_rng(1)
    q = AffineBody.rest_q([0.1, 0.2, 0.3]) + 0.05 * rng.standard_normal((4, 3))
    e0, g = b.ortho(q)
    h = 1e-6
    for a in range(4):
        for i in range(3):
            dq = np.zeros((4, 3)); dq[a, i] = h
            fd = (b.ortho(q + dq)[0] - b.ortho(q - dq)[0]) / (2 * h)
            assert abs(fd - g[a, i]) <= 1e-6 * max(1.0, abs(g).max()), (a, i)
    # on a rotation (r = 0) the Gauss-Newton operator IS the Hessian; everywhere it is symmetric PSD and the rotations are its null space there
    th = 0.7
    Rm = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    qr = np.concatenate([[[0.0, 0, 0]], Rm.T], 0)  # columns of A = rows here
    assert b.ortho(qr)[0] < 1e-20
    for _ in range(5):
        p = rng.standard_normal((4, 3))
        fd = (b.ortho(qr + 1e-6 * p)[1] - b.ortho(qr - 1e-6 * p)[1]) / 2e-6
        assert np.abs(fd - b.ortho_hess_vec(qr, p)).max() <= 1e-5 * np.abs(fd).max()
        p2 = rng.standard_normal((4, 3))
        assert abs((p * b.ortho_hess_vec(q, p2)).sum() - (p2 * b.ortho_hess_vec(q, p)).sum()) <= 1e-9 * b.kv
        assert (p * b.ortho_hess_vec(q, p)).sum() >= 0.0
    D = b.ortho_diag_blocks(q)
    for m in range(1, 4):
        for i in range(3):
            e = np.zeros((4, 3)); e[m, i] = 1.0
            assert np.allclose(b.ortho_hess_vec(q, e)[m], D[m][:, i])


def test_scene_gradient_matches_finite_differences_with_pairs_both_ways_and_ground():
    sc, y, cons = _scene()
    rng = np.random.default_rng(2)
    y = y + 2e-5 * rng.standard_normal(y.shape)
    kinds = sc.pairs(y)
    assert len(kinds[0][0]) >= 1 and len(kinds[1][0]) >= 1, [len(k[0]) for k in kinds]  # pad vertex / ball triangle AND ball vertex / pad triangle
    assert (sc.ball.points(y[sc.V:])[:, 2] - sc.gh < sc.dhat).any()  # the ground barrier acts on the ball
    yt = y + 1e-5 * rng.standard_normal(y.shape)
    aim = y[:sc.V] + 1e-5
    g = sc.gradient(y, yt, cons, aim)
    h = 1e-9
    rows = list(kinds[0][0][:2]) + list(sc.pad_tris[kinds[1][1][0]]) + [sc.V, sc.V + 1, sc.V + 2, sc.V + 3] + [0, 7]
    for r in rows:
        for i in range(3):
            e = np.zeros_like(y); e[r, i] = h
            fd = (sc.energy(y + e, yt, cons, aim) - sc.energy(y - e, yt, cons, aim)) / (2 * h)
            assert abs(fd - g[r, i]) <= 2e-4 * max(abs(g[r, i]), 1e-3 * np.abs(g).max()), (r, i, fd, g[r, i])


def test_scene_operator_is_symmetric_psd_and_diag_blocks_are_its_diagonal():
    sc, y, cons = _scene()
    rng = np.random.default_rng(3)
    p, p2 = rng.standard_normal(y.shape), rng.standard_normal(y.shape)
    Hp, Hp2 = sc.hess_vec(y, p, cons), sc.hess_vec(y, p2, cons)
    assert abs((p2 * Hp).sum() - (p * Hp2).sum()) <= 1e-10 * abs((p * Hp).sum())
    assert (p * Hp).sum() > 0
    B = sc.ball_block(y)  # the exactly inverted part of the preconditioner = the operator restricted to the ball's rows
    for a in range(4):
        for i in range(3):
            e = np.zeros_like(y); e[sc.V + a, i] = 1.0
            assert np.allclose(sc.hess_vec(y, e, cons)[sc.V:].reshape(12), B[:, 3 * a + i], rtol=1e-9, atol=1e-16)
    assert np.linalg.eigvalsh(B).min() > 0
    D = sc.diag_blocks(y, cons)
    kinds = sc.pairs(y)
    for r in [int(kinds[0][0][0]), int(sc.pad_tris[kinds[1][1][0]][0]), sc.V, sc.V + 1, sc.V + 3]:
        for i in range(3):
            e = np.zeros_like(y); e[r, i] = 1.0
            assert np.allclose(sc.hess_vec(y, e, cons)[r], D[r][:, i], rtol=1e-10, atol=1e-14), (r, i)


def test_pair_forces_are_action_and_reaction():
    """The pair energy is invariant under a common translation of pad and ball: the pair forces on the pad vertices and on the ball's
    translation p sum to zero (the rows of every pair's distance gradient sum to zero)."""
    sc, y, cons = _scene()
    rows, coef, w, d, n, mol, _ = sc._pair_rows(y)
    assert (mol[len(w) - len(sc.pairs(y)[2][0]):] <= 1.0).all() and len(sc.pairs(y)[2][0]) > 0  # edge-edge pairs are in the list
    tr = np.zeros(len(w))
    for r in range(8):
        on_p_or_pad = (rows[:, r] < sc.V) | (rows[:, r] == sc.V)
        tr += np.where(on_p_or_pad, coef[:, r], 0.0)
    assert np.abs(tr).max() < 1e-12


def test_free_fall_is_exact_and_ball_rests_on_the_ground_with_its_weight():
    sc, y, cons = _scene(press=-5e-3)  # pad far above: no pairs
    V = sc.V
    y[V, 2] += 0.002  # ball lifted out of every barrier zone (the pad's face is 5.5 mm above it)
    v = np.zeros_like(y)
    y1, v1, info = sc.step(y, v, cons, y[:V].copy(), velocity_tol=1e-6)
    g = 9.8
    assert np.allclose(y1[V], y[V] + np.array([0, 0, -g * sc.dt**2]), atol=1e-12) and np.allclose(y1[V + 1:], np.eye(3), atol=1e-12)
    # drop it back to the ground and let it settle: at rest the ground barrier carries the ball's weight
    y[V, 2] -= 0.002
    v = np.zeros_like(y)
    for _ in range(60):
        y, v, info = sc.step(y, v, cons, y[:V].copy(), velocity_tol=1e-5)
        v[V:] *= 0.5  # (numerical damping of the test: the barrier is a stiff undamped spring)
    xb = sc.ball.points(y[V:])
    f = sc._ground(xb, sc.ball.area)[1].sum() / sc.dt**2  # dE/dz summed = -force [N]
    weight = sc.ball.S[0, 0] * g
    assert xb[:, 2].min() > sc.gh
    assert abs(-f - weight) <= 0.02 * weight, (f, weight)


def test_pressed_pad_converges_to_a_stationary_point_without_penetration():
    """Three steps of the held back face moving down onto the ball, approached from outside every barrier zone: each Newton iteration
    decreases the potential, no pair and no ground gap closes, and the loop converges (quadratically at the end) to a state whose
    plain-potential gradient is below 1e-6 of the largest pair force.  The ball of THIS test is 100 x denser than the reference's: at
    1e3 kg/m^3 its rotational inertia is 1e-8 kg m^2, the lateral force a deforming pad triangle puts on it (1e-3 of the normal force)
    turns it by 0.1 rad per step, and an affine body takes a finite rotation in Newton steps of milliradians, bounded by its quartic
    orthogonality energy (Lan et al. 2022) - the loop then crawls for dozens of iterations, which says nothing about the terms."""
    from oracle.fem_oracle import barrier

    sc, y, cons = _scene(press=-2e-5, pole=True, shift=(0.0008, 0.0005), ground_gap=1.02)
    sc.ball = AffineBody(sc.ball.X, sc.ball.tris, density=1e5)
    V = sc.V
    aim = y[:V].copy()
    v = np.zeros_like(y)
    for k in range(3):
        aim[:, 2] -= 5e-5
        yt = y + sc.dt * v
        yt[:V, 2] -= 9.8 * sc.dt**2
        yt[V, 2] -= 9.8 * sc.dt**2
        y0 = y
        for it in range(25):
            y, st = sc.newton_step(y, yt, cons, aim, pcg_max_iter=3000, pcg_tol_rate=1e-12)
            assert st[1] <= st[0] and st[2] > 0.0
            if st[4] < 1e-9 and st[5] < 1e-7:
                break
        assert it < 24, (k, st)
        (pi, pj, pw, pd, pn, pb), (bi, bj, bw, bd, bn, bb) = sc.pairs(y)[:2]
        w, d = np.concatenate([pw, bw]), np.concatenate([pd, bd])
        assert len(d) > 0 and d.min() > 0
        scale = sc.dt**2 * sc.kappa * np.abs(w * barrier(d / sc.dhat)[1] / sc.dhat).max()
        assert np.abs(sc.gradient(y, yt, cons, aim)).max() <= 1e-6 * scale, (k, scale)
        assert (sc.ball.points(y[V:])[:, 2] > sc.gh).all()
        v = (y - y0) / sc.dt
    assert y[V, 2] < y0[V, 2]  # the ball is being pushed towards the ground


def test_additive_ccd_known_answers():
    from oracle.abd_oracle import accd_point_triangle

    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.0]])
    p = np.array([0.2, 0.2, 0.5])
    t = accd_point_triangle(p, tri, np.array([0, 0, -1.0]), np.zeros((3, 3)))  # head-on: stops with 10 % of the gap left
    assert abs(t - 0.45) < 1e-9
    assert accd_point_triangle(p, tri, np.array([0.3, 0, 0.0]), np.zeros((3, 3))) == 1.0  # sliding: the whole step
    assert accd_point_triangle(p, tri, np.array([0, 0, -1.0]), np.tile([0, 0, -1.0], (3, 1))) == 1.0  # common translation
    t = accd_point_triangle(p, tri, np.array([0, 0, -0.6]), np.tile([0, 0, 0.4], (3, 1)))  # both move: closing speed 1
    assert abs(t - 0.45) < 1e-9


def test_lagged_friction_terms_match_finite_differences_and_blocks():
    """Coulomb friction of every contact (pairs of both kinds and the ground under the ball), lagged at the step's start (IPC): FD gradient
    of the scene with friction on, the friction part of the operator against the FD derivative of the friction part of the gradient (the
    lagged potential's Hessian is exact), symmetry, PSD, and its share of the diagonal / ball blocks; in the stick AND in the slip regime."""
    sc, y0, cons = _scene()
    rng = np.random.default_rng(7)
    y0 = y0 + 2e-5 * rng.standard_normal(y0.shape)
    sc.mu = 0.5
    sc._lag = sc.friction_lag(y0)
    rows, coef, n, lam, _ = sc._lag
    assert len(lam) >= 3 and (lam > 0).all() and (rows[:, 0] >= sc.V).any() and (rows[:, 0] < sc.V).any()
    for scale in (2e-5, 5e-4):  # |u| below / above the stick tolerance eps_v * dt = 1e-4 m
        y = y0 + scale * rng.standard_normal(y0.shape) * (np.arange(len(y0))[:, None] >= 0)
        y[sc.V + 1:] = y0[sc.V + 1:] + 0.02 * scale / 5e-4 * rng.standard_normal((3, 3))
        f = sc._fric(y)
        yy = np.linalg.norm(f[3], axis=1)
        assert (yy < 1e-4).any() if scale < 1e-4 else (yy > 1e-4).any()
        yt = y0.copy()

        def parts(yv):
            g1 = sc.gradient(yv, yt, cons, y0[:sc.V])
            mu, sc.mu = sc.mu, 0.0
            g0 = sc.gradient(yv, yt, cons, y0[:sc.V])
            sc.mu = mu
            return g1 - g0

        def e_f(yv):
            e1 = sc.energy(yv, yt, cons, y0[:sc.V])
            mu, sc.mu = sc.mu, 0.0
            e0 = sc.energy(yv, yt, cons, y0[:sc.V])
            sc.mu = mu
            return e1 - e0

        gf = parts(y)
        h = 1e-8
        for r in [int(rows[0, 0]), int(rows[-1, 0]), sc.V, sc.V + 2]:
            for i in range(3):
                e = np.zeros_like(y); e[r, i] = h
                fd = (e_f(y + e) - e_f(y - e)) / (2 * h)
                assert abs(fd - gf[r, i]) <= 1e-5 * max(np.abs(gf).max(), 1e-30), (scale, r, i, fd, gf[r, i])
        p = rng.standard_normal(y.shape)
        mu = sc.mu
        Hp1 = sc.hess_vec(y, p, cons); sc.mu = 0.0; Hp0 = sc.hess_vec(y, p, cons); sc.mu = mu
        fd = (parts(y + 1e-7 * p) - parts(y - 1e-7 * p)) / 2e-7
        assert np.abs((Hp1 - Hp0) - fd).max() <= 2e-4 * np.abs(fd).max(), (scale, np.abs((Hp1 - Hp0) - fd).max(), np.abs(fd).max())
        assert (p * (Hp1 - Hp0)).sum() >= 0.0
        D1 = sc.diag_blocks(y, cons); B1 = sc.ball_block(y); sc.mu = 0.0; D0 = sc.diag_blocks(y, cons); B0 = sc.ball_block(y); sc.mu = mu
        for r in [int(rows[0, 0]), sc.V + 1]:
            for i in range(3):
                e = np.zeros_like(y); e[r, i] = 1.0
                a = sc.hess_vec(y, e, cons); sc.mu = 0.0; b = sc.hess_vec(y, e, cons); sc.mu = mu
                assert np.allclose((a - b)[r], (D1 - D0)[r][:, i], rtol=1e-9, atol=1e-18)
        for a4 in range(4):
            for i in range(3):
                e = np.zeros_like(y); e[sc.V + a4, i] = 1.0
                a = sc.hess_vec(y, e, cons); sc.mu = 0.0; b = sc.hess_vec(y, e, cons); sc.mu = mu
                assert np.allclose((a - b)[sc.V:].reshape(12), (B1 - B0)[:, 3 * a4 + i], rtol=1e-9, atol=1e-18)


def test_friction_keeps_the_light_ball_from_rolling_away():
    """The reference's contact model has friction (ratio 0.5, US:103-124).  With it the reference-density ball under the pressing pad needs a
    few Newton iterations per step at the default tolerances; without it the faceted ball rolls onto a facet and the loop runs into its cap."""
    res = {}
    for mu in (0.5, 0.0):
        sc, y, cons = _scene(press=-2e-5, pole=True, shift=(0.0008, 0.0005), ground_gap=1.02)
        sc.mu = mu
        V = sc.V
        aim, v, its = y[:V].copy(), np.zeros_like(y), []
        for k in range(5 if mu > 0 else 4):
            aim[:, 2] -= 6e-5
            y, v, info = sc.step(y, v, cons, aim, max_newton=40)
            assert int(info[2]) == 0
            its.append(int(info[0]))
        res[mu] = (its, np.abs(y[V + 1:] - np.eye(3)).max())
    assert max(res[0.5][0]) <= 6 and res[0.5][1] < 0.02, res
    assert max(res[0.0][0]) == 40, res  # (the frictionless ball is what runs into the cap)


def test_segment_segment_regions_and_derivatives():
    """Distance of two segments in all nine parameter regions; its gradient on the four end points is the envelope formula."""
    from oracle.abd_oracle import segment_segment
    rng = np.random.default_rng(3)
    seen = set()
    h = 1e-7
    for _ in range(300):
        P = rng.uniform(-1.0, 1.0, (4, 3))
        s, t, d, n = segment_segment(P[0:1], P[1:2], P[2:3], P[3:4])
        s, t, d, n = s[0, 0], t[0, 0], d[0, 0], n[0, 0]
        # brute force over a grid of both parameters bounds the minimum from above
        ss, tt = np.meshgrid(np.linspace(0, 1, 41), np.linspace(0, 1, 41), indexing="ij")
        dd = np.linalg.norm((P[0] + ss[..., None] * (P[1] - P[0])) - (P[2] + tt[..., None] * (P[3] - P[2])), axis=-1)
        assert d <= dd.min() + 1e-12 and d >= dd.min() - 0.05
        seen.add((0 if s <= 0 else 2 if s >= 1 else 1, 0 if t <= 0 else 2 if t >= 1 else 1))
        if min(abs(s), abs(1 - s), abs(t), abs(1 - t)) < 1e-4 and 0 < s < 1 and 0 < t < 1:
            continue
        coef = [(1 - s), s, -(1 - t), -t]
        for k in range(4):
            for i in range(3):
                Pp, Pm = P.copy(), P.copy()
                Pp[k, i] += h; Pm[k, i] -= h
                fd = (segment_segment(Pp[0:1], Pp[1:2], Pp[2:3], Pp[3:4])[2][0, 0] - segment_segment(Pm[0:1], Pm[1:2], Pm[2:3], Pm[3:4])[2][0, 0]) / (2 * h)
                assert abs(fd - coef[k] * n[i]) < 2e-6, (k, i, s, t)
    assert len(seen) >= 8  # interior-interior, the four edges of the parameter square, corners


def test_parallel_segments_have_a_finite_distance():
    from oracle.abd_oracle import segment_segment
    a0, a1 = np.array([[0.0, 0, 0]]), np.array([[1.0, 0, 0]])
    b0, b1 = np.array([[0.2, 0, 0.3]]), np.array([[0.7, 0, 0.3]])
    s, t, d, n = segment_segment(a0, a1, b0, b1)
    assert d[0, 0] == pytest.approx(0.3) and np.allclose(n[0, 0], [0, 0, -1.0])


def test_edge_edge_pairs_energy_gradient_match_finite_differences():
    """The scene with edge-edge pairs: (i) they exist at the usual press; (ii) gradient = FD of the energy on rows an edge-edge pair touches;
    (iii) a pad edge turned parallel to a ball edge is mollified (m < 1) and the mollifier's own gradient term is in the gradient."""
    sc, y, cons = _scene(press=3.5e-4)
    ee = sc.pairs(y)[2]
    assert len(ee[0]) > 0 and (ee[7] <= 1.0).all()
    yt = y + 1e-5
    g = sc.gradient(y, yt, cons, y[: sc.V])
    rows = sorted({int(r) for r in sc.pad_edges[ee[0]].reshape(-1)})[:6] + [sc.V, sc.V + 2]
    h = 1e-9
    for r in rows:
        for i in range(3):
            yp, ym = y.copy(), y.copy()
            yp[r, i] += h; ym[r, i] -= h
            fd = (sc.energy(yp, yt, cons, y[: sc.V]) - sc.energy(ym, yt, cons, y[: sc.V])) / (2 * h)
            assert abs(fd - g[r, i]) <= 2e-5 * max(abs(g[r, i]), np.abs(g).max() * 1e-3), (r, i, fd, g[r, i])
    # (iii) lay one close pad edge nearly parallel over its partner ball edge, 0.4 d_hat away along the pair's normal
    k = int(np.argmin(ee[3]))
    pe, be = sc.pad_edges[ee[0][k]], sc.ball_edges[ee[1][k]]
    xb = sc.ball.points(y[sc.V:])
    e2 = xb[be[1]] - xb[be[0]]
    u2 = e2 / np.linalg.norm(e2)
    nn = ee[4][k] - (ee[4][k] @ u2) * u2
    nn /= np.linalg.norm(nn)
    mid = xb[be[0]] + 0.5 * e2 + 0.4 * sc.dhat * nn
    L = np.linalg.norm(y[pe[1]] - y[pe[0]])
    skew = 0.01 * L * np.cross(u2, nn)
    y2 = y.copy()
    y2[pe[0]] = mid - 0.3 * L * u2 - skew
    y2[pe[1]] = mid + 0.3 * L * u2 + skew
    _, _, w, d, _, mol, X = sc._pair_rows(y2)
    assert X is not None and (mol < 1.0).any() and d.min() > 1e-5
    g2 = sc.gradient(y2, yt, cons, y[: sc.V])
    for r in (int(pe[0]), int(pe[1]), sc.V + 1):
        for i in range(3):
            yp, ym = y2.copy(), y2.copy()
            yp[r, i] += h; ym[r, i] -= h
            fd = (sc.energy(yp, yt, cons, y[: sc.V]) - sc.energy(ym, yt, cons, y[: sc.V])) / (2 * h)
            assert abs(fd - g2[r, i]) <= 1e-4 * max(abs(g2[r, i]), np.abs(g2).max() * 1e-3), (r, i, fd, g2[r, i])


def test_accd_edge_edge_is_conservative():
    from oracle.abd_oracle import accd_edge_edge, segment_segment
    ea = np.array([[0.0, 0, 0], [1.0, 0, 0]])
    eb = np.array([[0.5, -0.5, 0.2], [0.5, 0.5, 0.2]])
    dea = np.zeros((2, 3))
    deb = np.array([[0.0, 0, -1.0], [0.0, 0, -1.0]])  # b falls onto a: impact at t = 0.2
    t = accd_edge_edge(ea, eb, dea, deb, 1.0)
    assert 0.17 < t < 0.2
    d = segment_segment(ea[0:1], ea[1:2], (eb[0] + t * deb[0])[None], (eb[1] + t * deb[1])[None])[2][0, 0]
    assert d >= 0.1 * 0.2 - 1e-12
    assert accd_edge_edge(ea, eb, dea, np.array([[1.0, 0, 0], [1.0, 0, 0]]), 1.0) == 1.0  # sliding keeps the gap
