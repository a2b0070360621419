"""CPU oracle for the reference's own UIPC scene (TEST INFRASTRUCTURE - never imported by the product path): a FREE affine-body ball
lying on a ground plane under the gelpad, in IPC contact with it through vertex-triangle pairs in both directions.

PARITY UNPINNED, like oracle/fem_oracle.py and for the same reason: the arithmetic lives in libuipc (un-vendored submodule,
`.gitmodules:1-4`).  What the reference shows of this scene are its call sites:

  * scripts/benchmarking/tactile_sim_performance/envs/ball_rolling_uipc.py:71-92 - `UipcSimCfg(ground_height=0.001,
    contact=Contact(d_hat=0.0005))`, the ball as `UipcObjectCfg(constitution_cfg=AffineBodyConstitutionCfg())` at z = 0.01;
  * source/tacex_uipc/tacex_uipc/objects/uipc_object.py:62-74, 456-466 - `AffineBodyConstitution().apply_to(mesh, m_kappa * MPa,
    mass_density)`, m_kappa = 100 MPa, density 1e3, `kinematic=False`: the body's 12 degrees of freedom are free;
  * source/tacex_uipc/tacex_uipc/sim/uipc_sim.py:192-201 - `ground(ground_height, ground_normal)` and ONE default contact model
    (friction rate, resistance in GPa) for every pair of surfaces.

The model follows the PUBLISHED sources libuipc implements:

  * Lan, Kaufman, Li, Jiang, Yang 2022, "Affine Body Dynamics": a body is x(X) = A X + p with q = (p, A) in R^12, kinetic term
    1/2 (q - q~)^T M (q - q~) with the 12 x 12 mass matrix M = int rho J^T J dV, J = d x / d q, and the orthogonality energy
    kappa * vol * |A A^T - I|_F^2 (eq. 7 there).  Here q is held as FOUR 3-vectors (p, c_1, c_2, c_3), c_k = column k of A, so
    x = p + sum_k X_k c_k, J is block-scalar and M = S (x) I_3 with the 4 x 4 moment matrix S = int rho (1, X)(1, X)^T dV taken over the
    closed surface mesh (signed tetrahedra against the origin) - the layout the kernel uses: the body is four more "vertices" of the env.
  * Li et al. 2020 (IPC): barrier kappa * w * b(d / d_hat), b(s) = -(s - 1)^2 ln s, on every point-triangle pair closer than d_hat:
    pad surface vertex against ball triangle (weight = the pad vertex's surface area) AND ball vertex against pad surface triangle
    (weight = the ball vertex's area); the ground half-space against the surface vertices of both bodies.  Point-triangle distance by
    region (Ericson 5.1.5); in every region grad_p d = n and grad_{corner j} d = -beta_j n with beta the barycentric coordinates of the
    closest point.  Hessian: the Gauss-Newton part b'' grad d grad d^T (b' hess d is dropped: the PSD projection the analytic contact of
    fem_oracle.py uses) and, for the orthogonality energy, 4 kappa vol [delta_mn A A^T + c_n c_m^T] (the term r_mn I of the exact
    Hessian, r = A^T A - I, is dropped: it vanishes on rotations and is what makes the exact Hessian indefinite under compression).
    Gradients are exact, so Newton converges to stationary points of the plain potential (tests/test_abd_oracle.py checks them by
    finite differences).  Lagged Coulomb friction (Li et al. 2020 eq. 18-20, section 5.4: normal force, normal and barycentric weights
    of every contact frozen at the state the step starts from) acts on the pairs of both kinds and on the ground contacts of both
    bodies once `BallScene.mu` is set - the reference's one default contact model (US:103-124, 192-201: ratio 0.5, eps_velocity 0.01).
    EDGE-EDGE pairs (BallScene(edge_edge=True), the default) complete IPC's contact set: every pad surface edge against every ball edge closer
    than d_hat by the segment-segment distance (Ericson 5.1.9; gradient (1-s) n, s n, -(1-t) n, -t n on the four end points), weighted with
    the mean of the two edges' areas (an edge stands for a third of its triangles' rest areas) and IPC's mollifier m(c) = (2 - c/eps) c/eps
    below eps = 1e-3 |e_a|^2 |e_b|^2 (rest lengths), c = |e_a x e_b|^2 (Li et al. 2020 eq. 24), whose own gradient b m'(c) grad c is part of
    the gradient; Hessian m b'' grad d grad d^T; the pair slides with friction like the others; additive CCD with
    l = max |d a_k| + max |d b_k|.
"""
from __future__ import annotations

import numpy as np

from .fem_oracle import LS_RESCUE, FemModel, barrier, friction_f0, pcg_solve


def icosphere(radius: float, level: int = 2):
    """(vertices (nv,3), triangles (nt,3), outward oriented): 12 / 42 / 162 / 642 vertices at level 0 / 1 / 2 / 3."""
    t = (1.0 + 5.0**0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        mid, nf = {}, []

        def m(a, b):
            key = (min(a, b), max(a, b))
            if key not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[key] = len(v) - 1
            return mid[key]

        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v) * radius, np.asarray(f, np.int32)


def point_triangle(p, a, b, c):
    """Closest points of triangles (a, b, c: (Nt,3)) to points p (Np,3): barycentric coordinates (Np,Nt,3) of the closest point by
    Ericson's regions in the book's order (the kernel's order), distance (Np,Nt) and unit vector n (Np,Nt,3) from closest point to p."""
    p = np.asarray(p, np.float64)[:, None, :]
    A, B, C = a[None], b[None], c[None]
    ab, ac = B - A, C - A
    ap = p - A
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - B
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - C
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        t_ab, t_ac = d1 / (d1 - d3), d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        den = 1.0 / (va + vb + vc)
    s, u = vb * den, vc * den  # face region
    z, o = np.zeros_like(d1), np.ones_like(d1)
    regions = [((d1 <= 0) & (d2 <= 0), z, z), ((d3 >= 0) & (d4 <= d3), o, z), ((vc <= 0) & (d1 >= 0) & (d3 <= 0), t_ab, z),
               ((d6 >= 0) & (d5 <= d6), z, o), ((vb <= 0) & (d2 >= 0) & (d6 <= 0), z, t_ac),
               ((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), 1.0 - t_bc, t_bc)]
    for cond, sr, ur in reversed(regions):  # first matching region wins
        s, u = np.where(cond, sr, s), np.where(cond, ur, u)
    beta = np.stack([1.0 - s - u, s, u], -1)
    q = A + s[..., None] * ab + u[..., None] * ac
    r = p - q
    d = np.linalg.norm(r, axis=-1)
    return beta, d, r / np.maximum(d, 1e-300)[..., None]


def accd_point_triangle(p, tri, dp, dtri, t_max=1.0, slack=0.9, keep=0.1, max_iter=64):
    """Additive CCD of ONE point-triangle pair (Li et al. 2021, "Codimensional IPC", Algorithm 1; no thickness): the largest t <= t_max
    up to which the pair moving along (dp, dtri) provably keeps a gap >= keep * d(0).  p (3,), tri (3,3), dp (3,), dtri (3,3)."""
    mean = (dp + dtri.sum(0)) / 4.0
    dp, dtri = dp - mean, dtri - mean
    l = np.linalg.norm(dp) + np.linalg.norm(dtri, axis=1).max()
    if not l > 0.0:
        return t_max
    dist = lambda t: float(point_triangle((p + t * dp)[None], *((tri[k] + t * dtri[k])[None] for k in range(3)))[1][0, 0])
    d0 = dist(0.0)
    g = keep * d0
    t, tl = 0.0, (1.0 - keep) * d0 / l
    for _ in range(max_iter):
        d = dist(t + tl)
        if t > 0.0 and d < g:
            break
        t += tl
        if t >= t_max:
            return t_max
        tl = slack * d / l
    return t


def segment_segment(a0, a1, b0, b1):
    """Closest points of segments a (Na,3)+(Na,3) and b (Nb,3)+(Nb,3), every a against every b (Ericson, Real-Time Collision Detection
    5.1.9, in the book's order - the kernel's order): parameters s, t in [0,1] (Na,Nb), distance (Na,Nb), unit vector n (Na,Nb,3) from the
    point on b to the point on a.  The gradient of the distance is (1-s) n, s n on a's end points and -(1-t) n, -t n on b's (envelope)."""
    a0, a1, b0, b1 = (np.asarray(v, np.float64) for v in (a0, a1, b0, b1))
    d1, d2 = (a1 - a0)[:, None, :], (b1 - b0)[None, :, :]
    r = a0[:, None, :] - b0[None, :, :]
    a, e = (d1 * d1).sum(-1), (d2 * d2).sum(-1)
    f, c, b = (d2 * r).sum(-1), (d1 * r).sum(-1), (d1 * d2).sum(-1)
    a, e = np.broadcast_to(a, b.shape), np.broadcast_to(e, b.shape)
    den = a * e - b * b
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(den > 0.0, np.clip((b * f - c * e) / den, 0.0, 1.0), 0.0)
        t = (b * s + f) / e
        s = np.where(t < 0.0, np.clip(-c / a, 0.0, 1.0), np.where(t > 1.0, np.clip((b - c) / a, 0.0, 1.0), s))
    t = np.clip(t, 0.0, 1.0)
    w = r + s[..., None] * d1 - t[..., None] * d2
    d = np.linalg.norm(w, axis=-1)
    return s, t, d, w / np.maximum(d, 1e-300)[..., None]


def edge_mollifier(c, eps):
    """IPC's mollifier of nearly parallel edge pairs (Li et al. 2020, eq. 24) in c = |e_a x e_b|^2: m = (2 - c/eps) c/eps below eps, 1 above;
    returns (m, dm/dc)."""
    r = c / eps
    lo = r < 1.0
    return np.where(lo, (2.0 - r) * r, 1.0), np.where(lo, 2.0 * (1.0 - r) / eps, 0.0)


def accd_edge_edge(ea, eb, dea, deb, t_max=1.0, slack=0.9, keep=0.1, max_iter=64):
    """Additive CCD of ONE edge-edge pair (as accd_point_triangle; l = max |d a_k| + max |d b_k|).  ea, eb, dea, deb (2,3)."""
    mean = (dea.sum(0) + deb.sum(0)) / 4.0
    dea, deb = dea - mean, deb - mean
    l = np.linalg.norm(dea, axis=1).max() + np.linalg.norm(deb, axis=1).max()
    if not l > 0.0:
        return t_max
    dist = lambda t: float(segment_segment((ea[0] + t * dea[0])[None], (ea[1] + t * dea[1])[None], (eb[0] + t * deb[0])[None], (eb[1] + t * deb[1])[None])[2][0, 0])
    d0 = dist(0.0)
    g = keep * d0
    t, tl = 0.0, (1.0 - keep) * d0 / l
    for _ in range(max_iter):
        d = dist(t + tl)
        if t > 0.0 and d < g:
            break
        t += tl
        if t >= t_max:
            return t_max
        tl = slack * d / l
    return t


def surface_edges(tris, X):
    """Unique edges (E,2) (lower index first, sorted) of a triangle mesh, the area each stands for (a third of its triangles' rest areas - the
    edge areas sum to the surface area) and its squared rest length."""
    tris = np.asarray(tris, np.int64)
    ta = 0.5 * np.linalg.norm(np.cross(X[tris[:, 1]] - X[tris[:, 0]], X[tris[:, 2]] - X[tris[:, 0]]), axis=1)
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    e.sort(axis=1)
    key = e[:, 0] * (int(tris.max()) + 1) + e[:, 1]
    uk, inv = np.unique(key, return_inverse=True)
    area = np.zeros(len(uk))
    np.add.at(area, inv, np.tile(ta / 3.0, 3))
    edges = np.stack([uk // (int(tris.max()) + 1), uk % (int(tris.max()) + 1)], 1)
    return edges, area, ((X[edges[:, 1]] - X[edges[:, 0]]) ** 2).sum(-1)


class AffineBody:
    """One affine body on its closed surface mesh (body frame: the mesh's own coordinates)."""

    def __init__(self, verts, tris, density=1e3, kappa=100e6):
        self.X = np.asarray(verts, np.float64)
        self.tris = np.asarray(tris, np.int64)
        a, b, c = (self.X[self.tris[:, k]] for k in range(3))
        det = np.einsum("ij,ij->i", a, np.cross(b, c))
        assert det.sum() > 0, "surface triangles must be oriented outward"
        vol = det / 6.0
        self.vol = float(vol.sum())
        S = np.zeros((4, 4))
        S[0, 0] = self.vol
        S[0, 1:] = S[1:, 0] = (vol[:, None] * (a + b + c) / 4.0).sum(0)
        sm = a + b + c
        S[1:, 1:] = (vol[:, None, None] / 20.0 * (a[:, :, None] * a[:, None] + b[:, :, None] * b[:, None] + c[:, :, None] * c[:, None]
                                                   + sm[:, :, None] * sm[:, None])).sum(0)
        self.S = density * S          # 4 x 4 moment matrix: M = S (x) I_3
        self.kv = kappa * self.vol    # orthogonality stiffness kappa * vol [J]
        ta = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
        self.area = np.zeros(len(self.X))
        np.add.at(self.area, self.tris.reshape(-1), np.repeat(ta / 3.0, 3))
        self.Y = np.concatenate([np.ones((len(self.X), 1)), self.X], 1)  # (nv,4): x = Y q

    @staticmethod
    def rest_q(centre):
        return np.concatenate([np.asarray(centre, np.float64)[None], np.eye(3)], 0)

    def points(self, q):
        return self.Y @ q  # (nv,3)

    def to_q(self, f):
        """Forces on the surface vertices (nv,3) -> generalised force (4,3)."""
        return self.Y.T @ f

    def ortho(self, q):
        """kappa vol |A^T A - I|^2, its gradient (4,3) and r = A^T A - I."""
        c = q[1:]
        r = c @ c.T - np.eye(3)
        g = np.zeros((4, 3))
        g[1:] = 4.0 * self.kv * (r @ c)
        return self.kv * (r * r).sum(), g

    def ortho_hess_vec(self, q, pq):
        """Gauss-Newton part 4 kappa vol [delta_mn A A^T + c_n c_m^T] applied to pq (4,3)."""
        c, pc = q[1:], pq[1:]
        out = np.zeros((4, 3))
        out[1:] = 4.0 * self.kv * (pc @ (c.T @ c) + (c @ pc.T) @ c)  # row m: A A^T p_m + sum_n c_n (c_m . p_n)
        return out

    def ortho_diag_blocks(self, q):
        c = q[1:]
        AAt = c.T @ c
        D = np.zeros((4, 3, 3))
        for m in range(3):
            D[1 + m] = 4.0 * self.kv * (AAt + np.outer(c[m], c[m]))
        return D


class BallScene:
    """Gelpad (FemModel) + free affine-body ball + ground half-space z >= ground_height + IPC pairs between pad and ball.

    State of one env: y (V + 4, 3) = pad vertices, then the ball's (p, c_1, c_2, c_3).  All energies are those of ONE backward-Euler
    step (already scaled by dt^2 where they are potentials)."""

    def __init__(self, pad: FemModel, pad_tris, pad_area, ball: AffineBody, dhat=5e-4, kappa=None, ground_height=0.001, resistance=10.0,
                 edge_edge=True):
        self.pad, self.ball = pad, ball
        self.edge_edge = bool(edge_edge)
        self.pad_tris = np.asarray(pad_tris, np.int64)
        self.pad_area = np.asarray(pad_area, np.float64)
        self.dhat = float(dhat)
        self.kappa = resistance * 1e9 * dhat if kappa is None else float(kappa)  # as UipcSim: resistance [GPa] * d_hat [J/m^2]
        self.gh = float(ground_height)
        self.dt = pad.dt
        self.V = len(pad.X)
        self.pad_sv = np.where(self.pad_area > 0)[0]
        # edge-edge pairs (pad surface edge x ball edge): weight = the mean of the two edge areas; mollifier threshold 1e-3 |e_a|^2 |e_b|^2 (rest)
        self.pad_edges, self.pad_earea, self.pad_elen2 = surface_edges(self.pad_tris, pad.X)
        self.ball_edges, self.ball_earea, self.ball_elen2 = surface_edges(ball.tris, ball.X)
        # lagged Coulomb friction of every contact (Li et al. 2020, eq. 18-20; US:103-124 friction ratio 0.5, eps_velocity 0.01): off until
        # `mu` is set; the lag (normal force, normal, the pair's coefficients) is taken by step() at the state the step starts from
        self.mu, self.eps_v = 0.0, 0.01
        self._lag = None

    # ---- contact terms -----------------------------------------------------------------------------------------------------------
    def _ground(self, x, w):
        """Ground barrier of points x (N,3) with weights w: energy, forces (N,3) = gradient, curvature b'' (N,) along z."""
        d = x[:, 2] - self.gh
        b, b1, b2 = barrier(d / self.dhat)
        on = w > 0
        with np.errstate(invalid="ignore"):
            e = np.where(on, w * b, 0.0).sum()
        k = self.dt**2 * self.kappa
        return k * e, k * w * b1 / self.dhat, k * w * b2 / self.dhat**2, np.where(on, d, np.inf)

    def pairs(self, y):
        """Every point-triangle pair closer than d_hat as [kind 0, kind 1] = (point index, triangle row, weight, d, n (3,), beta (3,)) arrays:
        kind 0 = pad vertex vs ball triangle, kind 1 = ball vertex vs pad triangle; and, with edge_edge, the edge-edge pairs (below)."""
        xb = self.ball.points(y[self.V:])
        out = []
        bt, pt = self.ball.tris, self.pad_tris
        beta, d, n = point_triangle(y[self.pad_sv], xb[bt[:, 0]], xb[bt[:, 1]], xb[bt[:, 2]])
        i, j = np.where(d < self.dhat)
        out.append((self.pad_sv[i], j, self.pad_area[self.pad_sv[i]], d[i, j], n[i, j], beta[i, j]))
        beta, d, n = point_triangle(xb, y[pt[:, 0]], y[pt[:, 1]], y[pt[:, 2]])
        i, j = np.where(d < self.dhat)
        out.append((i, j, self.ball.area[i], d[i, j], n[i, j], beta[i, j]))
        if self.edge_edge:  # third entry: (pad edge, ball edge, weight, d, n, s, t, m, m' grad c on (a0, a1, b0, b1) (K,4,3))
            pe, be = self.pad_edges, self.ball_edges
            a0, a1, b0, b1 = y[pe[:, 0]], y[pe[:, 1]], xb[be[:, 0]], xb[be[:, 1]]
            s, t, d, n = segment_segment(a0, a1, b0, b1)
            i, j = np.where(d < self.dhat)
            e1, e2 = (a1 - a0)[i], (b1 - b0)[j]
            u = np.cross(e1, e2)
            m, dm = edge_mollifier((u * u).sum(-1), 1e-3 * self.pad_elen2[i] * self.ball_elen2[j])
            ga, gb = 2.0 * np.cross(e2, u), 2.0 * np.cross(u, e1)
            dc = dm[:, None, None] * np.stack([-ga, ga, -gb, gb], 1)
            out.append((i, j, 0.5 * (self.pad_earea[i] + self.ball_earea[j]), d[i, j], n[i, j], s[i, j], t[i, j], m, dc))
        return out

    def _pair_rows(self, y):
        """The pairs as rank-one rows: for pair k a sparse gradient of its distance over the V + 4 state rows -
        g_k = sum_r coef[k, r] * n_k at state row rows[k, r] (r < 8; unused slots have coef 0).  Returns (rows, coef, w, d, n, m, X): m the
        mollifier of the pair (1 for point-triangle pairs), X = None or (pair index, rows (K,6), m'(c) grad c (K,6,3)) of the mollified pairs."""
        memo = getattr(self, "_rows_memo", None)  # (the PCG applies the operator hundreds of times at ONE state)
        if memo is not None and memo[2] == self.edge_edge and memo[0].shape == y.shape and np.array_equal(memo[0], y):
            return memo[1]
        V, Y, bt, pt = self.V, self.ball.Y, self.ball.tris, self.pad_tris
        pr = self.pairs(y)
        (pi, pj, pw, pd, pn, pb), (bi, bj, bw, bd, bn, bb) = pr[0], pr[1]
        # kind 0: +1 on the pad vertex, -(beta . Y[tri]) on the four ball rows
        r0 = np.concatenate([pi[:, None], np.broadcast_to(V + np.arange(4), (len(pi), 4)), np.zeros((len(pi), 3), np.int64)], 1)
        c0 = np.concatenate([np.ones((len(pi), 1)), -np.einsum("kj,kja->ka", pb, Y[bt[pj]]), np.zeros((len(pi), 3))], 1)
        # kind 1: +Y[b] on the four ball rows, -beta on the three pad vertices of the triangle
        r1 = np.concatenate([np.broadcast_to(V + np.arange(4), (len(bi), 4)), pt[bj], np.zeros((len(bi), 1), np.int64)], 1)
        c1 = np.concatenate([Y[bi], -bb, np.zeros((len(bi), 1))], 1)
        R, C, W, D, N = [r0, r1], [c0, c1], [pw, bw], [pd, bd], [pn, bn]
        M, X = [np.ones(len(pw) + len(bw))], None
        if self.edge_edge:
            ei, ej, ew, ed, en, es, et, em, dc = pr[2]
            pe, be = self.pad_edges[ei], self.ball_edges[ej]
            # kind 2: (1-s), s on the pad edge's end points, -((1-t) Y[b0] + t Y[b1]) on the four ball rows
            r2 = np.concatenate([pe, np.broadcast_to(V + np.arange(4), (len(ei), 4)), np.zeros((len(ei), 2), np.int64)], 1)
            c2 = np.concatenate([(1.0 - es)[:, None], es[:, None], -((1.0 - et)[:, None] * Y[be[:, 0]] + et[:, None] * Y[be[:, 1]]), np.zeros((len(ei), 2))], 1)
            R.append(r2); C.append(c2); W.append(ew); D.append(ed); N.append(en); M.append(em)
            lo = np.where(em < 1.0)[0]
            if len(lo):  # the mollifier's own gradient m'(c) grad c, over the same six state rows (index into the full pair list)
                gq = Y[be[lo, 0]][:, :, None] * dc[lo, 2][:, None, :] + Y[be[lo, 1]][:, :, None] * dc[lo, 3][:, None, :]  # (K,4,3)
                X = (len(pw) + len(bw) + lo, r2[lo, :6], np.concatenate([dc[lo, :2], gq], 1))
        out = (np.concatenate(R), np.concatenate(C), np.concatenate(W), np.concatenate(D), np.concatenate(N), np.concatenate(M), X)
        self._rows_memo = (y.copy(), out, self.edge_edge)
        return out

    # ---- lagged friction ---------------------------------------------------------------------------------------------------------
    def friction_lag(self, y0):
        """The contacts of the state y0 as friction constraints for the step that starts there: per contact the state rows and coefficients
        of its relative displacement (pairs: point minus closest point of the triangle, the barycentric weights frozen; ground: the vertex
        itself), the contact normal and the normal force lam = -kappa w b'(d / d_hat) / d_hat [N] - all frozen for the step (IPC's lag)."""
        V = self.V
        rows, coef, w, d, n, mol, _ = self._pair_rows(y0)
        lam = -self.kappa * w * mol * barrier(d / self.dhat)[1] / self.dhat
        R, C, N, Lm = [rows], [coef], [n], [lam]
        z = np.array([0.0, 0.0, 1.0])
        for (x, wt, ball) in ((y0[:V], self.pad_area, False), (self.ball.points(y0[V:]), self.ball.area, True)):
            gap = x[:, 2] - self.gh
            on = np.where((wt > 0) & (gap > 0) & (gap < self.dhat))[0]
            if len(on) == 0:
                continue
            r = np.zeros((len(on), 8), np.int64)
            c = np.zeros((len(on), 8))
            if ball:
                r[:, :4] = V + np.arange(4)
                c[:, :4] = self.ball.Y[on]
            else:
                r[:, 0] = on
                c[:, 0] = 1.0
            R.append(r); C.append(c); N.append(np.broadcast_to(z, (len(on), 3)).copy())
            Lm.append(-self.kappa * wt[on] * barrier(gap[on] / self.dhat)[1] / self.dhat)
        return (np.concatenate(R), np.concatenate(C), np.concatenate(N), np.concatenate(Lm), y0.copy())

    def _fric(self, y):
        """(rows, coef, lam, u (K,3) tangential relative displacement since the step's start, n) of the lagged contacts; None when off."""
        if self._lag is None or not self.mu > 0.0 or len(self._lag[3]) == 0:
            return None
        rows, coef, n, lam, y0 = self._lag
        rel = np.zeros((len(lam), 3))
        for r in range(8):
            rel += coef[:, r, None] * (y[rows[:, r]] - y0[rows[:, r]])
        u = rel - (rel * n).sum(-1, keepdims=True) * n
        return rows, coef, lam, u, n

    def _fric_blocks(self, f):
        """(K,3,3) friction Hessian of every lagged contact in its relative displacement: dt^2 mu lam [a T + (b - a) t t^T]."""
        rows, coef, lam, u, n = f
        yy = np.linalg.norm(u, axis=1)
        _, a, b = friction_f0(yy, self.eps_v * self.dt)
        t = u / np.maximum(yy, 1e-300)[:, None]
        T = np.eye(3)[None] - n[:, :, None] * n[:, None, :]
        return (self.dt**2 * self.mu * lam)[:, None, None] * (a[:, None, None] * T + (b - a)[:, None, None] * t[:, :, None] * t[:, None, :])

    # ---- incremental potential -------------------------------------------------------------------------------------------------------
    def energy(self, y, yt, cons=None, aim=None):
        V = self.V
        x, q = y[:V], y[V:]
        dq = q - yt[V:]
        e = self.pad.energy(x, yt[:V], cons, aim) + 0.5 * np.einsum("ab,ai,bi->", self.ball.S, dq, dq) + self.dt**2 * self.ball.ortho(q)[0]
        e += self._ground(x, self.pad_area)[0] + self._ground(self.ball.points(q), self.ball.area)[0]
        k = self.dt**2 * self.kappa
        _, _, w, d, _, mol, _ = self._pair_rows(y)
        e += k * (w * mol * barrier(d / self.dhat)[0]).sum()
        f = self._fric(y)
        if f is not None:
            e += self.dt**2 * self.mu * (f[2] * friction_f0(np.linalg.norm(f[3], axis=1), self.eps_v * self.dt)[0]).sum()
        return e

    def gradient(self, y, yt, cons=None, aim=None):
        V = self.V
        x, q = y[:V], y[V:]
        g = np.zeros_like(y)
        g[:V] = self.pad.gradient(x, yt[:V], cons, aim)
        g[V:] = self.ball.S @ (q - yt[V:]) + self.dt**2 * self.ball.ortho(q)[1]
        g[:V, 2] += self._ground(x, self.pad_area)[1]
        fb = np.zeros((len(self.ball.X), 3))
        fb[:, 2] = self._ground(self.ball.points(q), self.ball.area)[1]
        g[V:] += self.ball.to_q(fb)
        rows, coef, w, d, n, mol, X = self._pair_rows(y)
        s = self.dt**2 * self.kappa * w * mol * barrier(d / self.dhat)[1] / self.dhat
        for r in range(8):
            np.add.at(g, rows[:, r], (s * coef[:, r])[:, None] * n)
        if X is not None:
            k, xr, xv = X
            sb = self.dt**2 * self.kappa * w[k] * barrier(d[k] / self.dhat)[0]
            for r in range(6):
                np.add.at(g, xr[:, r], sb[:, None] * xv[:, r])
        f = self._fric(y)
        if f is not None:
            frows, fcoef, lam, u, _ = f
            a = friction_f0(np.linalg.norm(u, axis=1), self.eps_v * self.dt)[1]
            for r in range(8):
                np.add.at(g, frows[:, r], (self.dt**2 * self.mu * lam * a * fcoef[:, r])[:, None] * u)
        return g

    def hess_vec(self, y, p, cons=None):
        V = self.V
        x, q = y[:V], y[V:]
        out = np.zeros_like(y)
        out[:V] = self.pad.hess_vec(x, p[:V], cons)
        out[V:] = self.ball.S @ p[V:] + self.dt**2 * self.ball.ortho_hess_vec(q, p[V:])
        out[:V, 2] += self._ground(x, self.pad_area)[2] * p[:V, 2]
        cb = self._ground(self.ball.points(q), self.ball.area)[2]
        fb = np.zeros((len(self.ball.X), 3))
        fb[:, 2] = cb * (self.ball.Y @ p[V:])[:, 2]
        out[V:] += self.ball.to_q(fb)
        rows, coef, w, d, n, mol, _ = self._pair_rows(y)
        wk = self.dt**2 * self.kappa * w * mol * barrier(d / self.dhat)[2] / self.dhat**2
        gp = np.zeros(len(w))
        for r in range(8):
            gp += coef[:, r] * (n * p[rows[:, r]]).sum(-1)
        for r in range(8):
            np.add.at(out, rows[:, r], (wk * gp * coef[:, r])[:, None] * n)
        f = self._fric(y)
        if f is not None:
            frows, fcoef = f[0], f[1]
            M = self._fric_blocks(f)
            wv = np.zeros((len(f[2]), 3))
            for r in range(8):
                wv += fcoef[:, r, None] * p[frows[:, r]]
            Mw = np.einsum("kij,kj->ki", M, wv)
            for r in range(8):
                np.add.at(out, frows[:, r], fcoef[:, r, None] * Mw)
        return out

    def diag_blocks(self, y, cons=None):
        """(V + 4, 3, 3) diagonal blocks of the operator hess_vec applies (block-Jacobi preconditioner)."""
        V = self.V
        x, q = y[:V], y[V:]
        D = np.zeros((V + 4, 3, 3))
        D[:V] = self.pad.diag_blocks(x, cons)
        D[V:] = np.diag(self.ball.S)[:, None, None] * np.eye(3) + self.dt**2 * self.ball.ortho_diag_blocks(q)
        D[:V, 2, 2] += self._ground(x, self.pad_area)[2]
        cb = self._ground(self.ball.points(q), self.ball.area)[2]
        D[V:, 2, 2] += (cb[:, None] * self.ball.Y**2).sum(0)
        rows, coef, w, d, n, mol, _ = self._pair_rows(y)
        wk = self.dt**2 * self.kappa * w * mol * barrier(d / self.dhat)[2] / self.dhat**2
        nn = n[:, :, None] * n[:, None, :]
        for r in range(8):
            np.add.at(D, rows[:, r], (wk * coef[:, r] ** 2)[:, None, None] * nn)
        f = self._fric(y)
        if f is not None:
            M = self._fric_blocks(f)
            for r in range(8):
                np.add.at(D, f[0][:, r], (f[1][:, r] ** 2)[:, None, None] * M)
        return D

    def ball_block(self, y):
        """(12,12) block of the operator on the ball's rows (row-major over (row a, axis i)): S (x) I + dt^2 Gauss-Newton orthogonality
        + the ground and pair curvatures restricted to the four ball rows.  The preconditioner inverts it exactly: the 3 x 3 diagonal
        blocks alone see the orthogonality stiffness (0.04 J) in every direction of every row, while the body's rotations - combinations
        ACROSS the rows - are held by its inertia alone (5e-9 kg m^2): block Jacobi leaves a condition number of 1e7 on twelve unknowns."""
        V = self.V
        q = y[V:]
        c = q[1:]
        B = np.kron(self.ball.S, np.eye(3))
        AAt = c.T @ c
        G = np.zeros((12, 12))
        for m in range(3):
            for n in range(3):
                blk = np.outer(c[n], c[m]) + (AAt if m == n else 0.0)
                G[3 + 3 * m:6 + 3 * m, 3 + 3 * n:6 + 3 * n] = blk
        B += self.dt**2 * 4.0 * self.ball.kv * G
        cb = self._ground(self.ball.points(q), self.ball.area)[2]
        YY = (cb[:, None, None] * self.ball.Y[:, :, None] * self.ball.Y[:, None, :]).sum(0)  # (4,4), acts on the z components
        for a in range(4):
            for b in range(4):
                B[3 * a + 2, 3 * b + 2] += YY[a, b]
        rows, coef, w, d, n, mol, _ = self._pair_rows(y)
        wk = self.dt**2 * self.kappa * w * mol * barrier(d / self.dhat)[2] / self.dhat**2
        cq = np.zeros((len(w), 4))
        for r in range(8):
            on = rows[:, r] >= V
            np.add.at(cq, (np.where(on)[0], rows[on, r] - V), coef[on, r])
        for k in range(len(w)):
            B += wk[k] * np.kron(np.outer(cq[k], cq[k]), np.outer(n[k], n[k]))
        f = self._fric(y)
        if f is not None:
            M = self._fric_blocks(f)
            fq = np.zeros((len(f[2]), 4))
            for r in range(8):
                on = f[0][:, r] >= V
                np.add.at(fq, (np.where(on)[0], f[0][on, r] - V), f[1][on, r])
            for k in range(len(f[2])):
                B += np.kron(np.outer(fq[k], fq[k]), M[k])
        return B

    def preconditioner(self, y, cons=None):
        """r -> M^-1 r: 3 x 3 block Jacobi on the pad vertices, the exact inverse of the 12 x 12 ball block on the ball's rows."""
        V = self.V
        Dinv = np.linalg.inv(self.diag_blocks(y, cons)[:V])
        Bc = np.linalg.cholesky(self.ball_block(y))

        def prec(r):
            z = np.empty_like(r)
            z[:V] = np.einsum("vij,vj->vi", Dinv, r[:V])
            z[V:] = np.linalg.solve(Bc.T, np.linalg.solve(Bc, r[V:].reshape(12))).reshape(4, 3)
            return z

        return prec

    # ---- step bound: additive CCD on the candidate pairs -----------------------------------------------------------------------
    def max_step(self, y, dy, slack=0.9, reach=2.0, keep=0.1, max_iter=64):
        """Largest step in (0, 1] that keeps every point-triangle pair and the ground at a positive gap.
          * ground: the gap of a surface vertex is linear in the step: slack * gap / descent;
          * pairs within reach * d_hat: ADDITIVE CCD (Li et al. 2021, Algorithm 1) per pair - with the pair's mean displacement removed,
            l = |dp| + max_j |dx_j| bounds the rate at which the (1-Lipschitz) distance can shrink, so t += slack * d(t) / l never
            passes the time of impact; the march stops when the gap falls below `keep` of its start value (a pair that only slides - a
            ball rolling under the pad - keeps its distance and reaches t = 1, where a one-shot bound d / l would cut the step to the
            ratio of gap and sliding distance);
          * pairs beyond cannot reach contact while no surface point moves further than slack * reach * d_hat / 2 in the step."""
        V = self.V
        xb = self.ball.points(y[V:])
        db = self.ball.Y @ dy[V:]
        nb = np.linalg.norm(db, axis=1)
        npad = np.linalg.norm(dy[:V], axis=1)
        a = 1.0
        for (x, dz, w) in ((y[:V], dy[:V, 2], self.pad_area), (xb, db[:, 2], self.ball.area)):
            gap = x[:, 2] - self.gh
            ok = (w > 0) & (dz < 0) & (gap > 0)
            if ok.any():
                a = min(a, float((slack * gap[ok] / -dz[ok]).min()))
        vmax = max(npad[self.pad_sv].max(initial=0.0), nb.max(initial=0.0))
        R = reach * self.dhat
        if vmax > 0:
            a = min(a, slack * R / (2.0 * vmax))
        bt, pt = self.ball.tris, self.pad_tris
        for (P, dP, T0, dT0, tri) in ((y[self.pad_sv], dy[self.pad_sv], xb, db, bt), (xb, db, y[:V], dy[:V], pt)):
            _, d, _ = point_triangle(P, T0[tri[:, 0]], T0[tri[:, 1]], T0[tri[:, 2]])
            for i, j in zip(*np.where(d < R)):
                a = min(a, accd_point_triangle(P[i], T0[tri[j]], dP[i], dT0[tri[j]], a, slack, keep, max_iter))
        if self.edge_edge:
            pe, be = self.pad_edges, self.ball_edges
            _, _, d, _ = segment_segment(y[pe[:, 0]], y[pe[:, 1]], xb[be[:, 0]], xb[be[:, 1]])
            for i, j in zip(*np.where(d < R)):
                a = min(a, accd_edge_edge(y[pe[i]], xb[be[j]], dy[pe[i]], db[be[j]], a, slack, keep, max_iter))
        return a

    # ---- one Newton iteration / one time step (the algorithm of fem_ball_newton_kernel) ------------------------------------------
    def newton_step(self, y, yt, cons=None, aim=None, pcg_max_iter=1024, pcg_tol_rate=1e-3, ls_max_iter=8, ls_refine=4):
        """PCG (block Jacobi on the pad + exact ball block) on the Gauss-Newton system, conservative step bound, backtracking line search (first E <= E0 wins,
        then `ls_refine` bisections towards the last rejected step when the step was cut).
        Returns (y_new, [E0, E1, step, pcg iterations, max |d| over the position rows, max |d| over the ball's affine rows])."""
        g = self.gradient(y, yt, cons, aim)
        d, it = pcg_solve(lambda p: self.hess_vec(y, p, cons), self.preconditioner(y, cons), -g, pcg_max_iter, pcg_tol_rate)
        E0 = self.energy(y, yt, cons, aim)
        step = self.max_step(y, d)
        step_full = step
        y_new, E1 = y, E0
        for _ in range(max(ls_max_iter, LS_RESCUE) + 1):
            cand = y + step * d
            Ec = self.energy(cand, yt, cons, aim)
            if Ec <= E0:
                y_new, E1 = cand, Ec
                break
            step *= 0.5
        else:
            step = 0.0
        if step > 0.0 and step < step_full:
            # a step that had to be cut: `ls_refine` bisections between it and the last rejected one keep the largest step that still does
            # not increase E - the pair that cut it ends INSIDE the barrier zone, in the next Hessian (csrc/fem_ball.h does the same)
            lo, hi = step, 2.0 * step
            for _ in range(ls_refine):
                mid = 0.5 * (lo + hi)
                Ec = self.energy(y + mid * d, yt, cons, aim)
                if Ec <= E0:
                    lo, y_new, E1 = mid, y + mid * d, Ec
                else:
                    hi = mid
            step = lo
        return y_new, np.array([E0, E1, step, it, np.abs(d[: self.V + 1]).max(), np.abs(d[self.V + 1:]).max()])

    def step(self, y, v, cons=None, aim=None, gravity=(0.0, 0.0, -9.8), max_newton=64, velocity_tol=0.05, transrate_tol=0.1, **kw):
        """One backward-Euler step: y~ = y + dt v + dt^2 g (gravity acts on the pad vertices and on the ball's translation p);
        Newton until the unscaled direction has max |d| <= velocity_tol * dt on every position row (pad vertices, the ball's p) AND
        max |d| <= transrate_tol * dt on the ball's affine rows (UipcSimCfg.newton: velocity_tol 0.05 m/s, transrate_tol 0.1 /s,
        uipc_sim.py:62-66 - libuipc's separate test for affine bodies).  Returns (y_new, v_new, [iterations, max |d| (positions), flags, pcg])."""
        dt = self.dt
        yt = y + dt * v
        g3 = dt * dt * np.asarray(gravity, np.float64)
        yt[: self.V] += g3
        yt[self.V] += g3
        y0, n, pcg, dmax, flags = y, 0, 0, np.inf, 0
        self._lag = self.friction_lag(y0) if self.mu > 0.0 else None
        for _ in range(max_newton):
            y, st = self.newton_step(y, yt, cons, aim, **kw)
            n += 1
            pcg += int(st[3])
            dmax = st[4]
            if dmax <= velocity_tol * dt and st[5] <= transrate_tol * dt:
                break
            if st[2] == 0.0:
                flags |= 2
                break
        return y, (y - y0) / dt, np.array([n, dmax, flags, pcg], np.float64)
