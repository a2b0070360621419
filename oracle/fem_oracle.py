"""CPU oracle for the gelpad FEM inner step (TEST INFRASTRUCTURE - never imported by the product path).

PARITY UNPINNED.  In the reference this arithmetic lives inside libuipc (C++/CUDA), an un-vendored git
submodule (`.gitmodules:1-4` -> github.com/DH-Ng/libuipc, branch `tacex`, commit not recoverable,
`source/tacex_uipc/libuipc/` is empty), reached through `world.advance()` (tacex_uipc/sim/uipc_sim.py:250-252)
for a gelpad declared as `StableNeoHookean` with `ElasticModuli.youngs_poisson(E, nu)` and a mass density
(tacex_uipc/objects/uipc_object.py:442-470, defaults E = 0.01 MPa, nu = 0.49, rho = 1e3 at :59,76-88),
driven by `SoftPositionConstraint` animation targets (tacex_uipc/sim/uipc_attachments.py:139-142,364-428).
The reference holds no test or golden vector at this boundary (tools/test_settings.py:54-57 skips libuipc's
tests; env_test_utils.py:62 excludes the Uipc envs), so this file restates the PUBLISHED model

    Smith, de Goes, Kim 2018, "Stable Neo-Hookean Flesh Simulation", eq. 14:
        Psi(F) = mu/2 (I_C - 3) + lambda/2 (J - alpha)^2 - mu/2 log(I_C + 1),   alpha = 1 + 3 mu / (4 lambda)
        with the paper's reparameterisation mu = 4/3 mu_Lame, lambda = lambda_Lame + 5/6 mu_Lame,

inside a backward-Euler incremental potential per environment

    E(x) = 1/2 sum_v m_v |x_v - xt_v|^2 + dt^2 sum_t vol_t (Psi(F_t) - Psi(I))
           + 1/2 s sum_{v in C} m_v |x_v - aim_v|^2          (soft position constraint, strength ratio s)

and pins ITSELF with known-answer tests (tests/test_fem_oracle.py): finite-difference consistency of
energy/gradient/Hessian, zero rest force, rigid-motion invariance, PSD projection, monotone line search.

Solver conventions restated from libuipc's / IPC's published sources (unpinned like everything here): the PCG's stopping test
`abs(r.z) <= tol_rate * abs(r0.z0)` of `LinearPCG::pcg` (libuipc, src/backends/cuda/linear_system/linear_pcg.cu; `pcg_solve`), the
Newton loop's test on the infinity norm of the search direction over dt (Li et al. 2020, Algorithm 1; `fem_step`), the friction lag
from the state the time step starts at (Li et al. 2020, section 5.4; `FrictionModel`, `fem_step(friction_lag="start")`).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import numpy as np


# --------------------------------------------------------------------------------------------------
# mesh I/O: Gmsh 2.2 ASCII (the reference ships real tet meshes under
# source/tacex_uipc/examples/libuipc-samples/tet_meshes/*.msh; fixtures under tests/golden/ are re-saved as .npz)
# --------------------------------------------------------------------------------------------------
def load_msh(path: Path):
    lines = Path(path).read_text().splitlines()
    i = lines.index("$Nodes")
    n = int(lines[i + 1])
    pts = np.array([[float(v) for v in lines[i + 2 + k].split()[1:4]] for k in range(n)])
    j = lines.index("$Elements")
    m = int(lines[j + 1])
    tets = []
    for k in range(m):
        p = lines[j + 2 + k].split()
        if int(p[1]) == 4:  # 4-node tetrahedron
            ntags = int(p[2])
            tets.append([int(v) - 1 for v in p[3 + ntags : 3 + ntags + 4]])
    return pts, np.array(tets, dtype=np.int32)


def box_tet_mesh(nx=6, ny=8, nz=3, size=(0.02075, 0.02525, 0.0045)):
    """Regular box split into 6 tets per cell: a gelpad-sized block (gsmini_cfg.py:22-24: 20.75 x 25.25 x 4.5 mm)."""
    xs, ys, zs = (np.linspace(0, size[d], n + 1) for d, n in enumerate((nx, ny, nz)))
    P = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)
    idx = lambda i, j, k: (i * (ny + 1) + j) * (nz + 1) + k
    tets = []
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                c = [idx(i + a, j + b, k + d) for a in (0, 1) for b in (0, 1) for d in (0, 1)]
                # Kuhn split around the main diagonal c[0]-c[7]
                for p in ((1, 3), (3, 2), (2, 6), (6, 4), (4, 5), (5, 1)):
                    tets.append([c[0], c[p[0]], c[p[1]], c[7]])
    return P, np.array(tets, dtype=np.int32)


# Backtracking beyond the configured cap (mirrors kLsRescue of csrc/fem_kernels.hip): when the capped line search finds no decrease
# the step is halved further, down to 2^-32, instead of leaving the env stuck - a Newton direction computed before a vertex enters
# the barrier zone knows nothing of the barrier it runs into.
LS_RESCUE = 32


# Safeguard of the coarse correction (kCoarseTrust of csrc/fem_kernels.hip): the PCG's stopping test is in the M^-1 norm, and M^-1
# contains the REST-state coarse operator - blind to the barrier / friction stiffness of the current contacts.  Where the coarse
# space holds nearly free modes (simple_axle held at its ends) the test passes with the residual's 2-norm above that of b; if it is
# above COARSE_TRUST |b| at exit (no reduction at all), the coarse part is dropped for the rest of the time step and the iteration's solve starts over.
COARSE_TRUST = 1.0


def pcg_solve(hv, prec, b, max_iter, tol_rate, d0=None, info=None):
    """Preconditioned CG on H d = b as the Newton kernels run it: stops when r^T M^-1 r <= tol_rate * b^T M^-1 b (or at max_iter) - libuipc's
    test (LinearPCG::pcg in src/backends/cuda/linear_system/linear_pcg.cu: `abs(rz_new) <= global_tol_rate * rz0`; relative on r.z itself,
    i.e. sqrt(tol_rate) on the M^-1 norm of the residual; source absent here, restated from the public repository) - and keeps
    what it has on negative curvature (first iteration from a zero start: the preconditioned steepest-descent direction).  `d0`: warm
    start (the part of the previous Newton direction that the CCD filter / the line search cut off).  Returns (d, iterations);
    `info` (a dict) receives "res_ratio" = |r|^2 / |b|^2 of the recurrence residual at exit (None after a negative-curvature exit)."""
    r = b.copy()
    z = prec(r)
    rz_b = (r * z).sum()
    warm = d0 is not None and np.abs(d0).max() > 0 and rz_b > 0
    d = d0.copy() if warm else np.zeros_like(b)
    if warm:
        r = r - hv(d)
        z = prec(r)
    p = z.copy()
    rz = (r * z).sum()
    it = 0
    neg_curv = False
    while it < max_iter and rz_b > 0 and rz > tol_rate * rz_b:
        Hp = hv(p)
        pHp = (p * Hp).sum()
        if pHp <= 0:
            if it == 0 and not warm:
                d = z.copy()
            neg_curv = True
            break
        al = rz / pHp
        d = d + al * p
        r = r - al * Hp
        z = prec(r)
        rz_new = (r * z).sum()
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
    if info is not None:
        info["neg_curv"] = neg_curv
        info["warm_discard"] = neg_curv and it == 0 and warm  # a warm start alone is no descent direction: the caller starts over from zero
        bb = (b * b).sum()
        info["res_ratio"] = None if neg_curv else ((r * r).sum() / bb if bb > 0 else 0.0)
    return d, it


def direction_cap(m, d):
    """Largest first step of a line search: no vertex moves by more than the rest mesh's bounding-box diagonal (a preconditioned
    steepest-descent direction taken on negative curvature has no length scale: M^-1 b with a soft coarse mode was 600 m on a 26 mm
    axle, out of reach of the 2^-40 the backtracking can do)."""
    dm = np.abs(d).max()
    cap = float(np.linalg.norm(np.ptp(m.X, axis=0)))
    return min(1.0, cap / dm) if dm > 0 else 1.0


def pcg_solve_guarded(hv, make_prec, b, max_iter, tol_rate, d0, coarse, state, model=None):
    """pcg_solve with the two safeguards of the Newton kernel.  `state` (a dict carried over the Newton iterations of ONE time step):
      "psd_safe"   - the PCG met negative curvature: `model` (the FemModel behind `hv`) switches to its PSD-safe Hessian (dpk1) for this
                     Newton iteration and the solve starts over (the next iteration tries the exact Hessian again);
      "coarse_off" - COARSE_TRUST fired: the coarse part of the preconditioner (`make_prec(coarse)` / `make_prec(None)`) is dropped for
                     the rest of the step and the solve starts over.
    Returns (d, iterations incl. those of discarded attempts)."""
    use = None if (coarse is None or (state is not None and state.get("coarse_off"))) else coarse
    info = {}
    if model is not None:
        model.psd_safe = False  # every Newton iteration tries the exact Hessian first: near the minimiser it is what converges quadratically
    d, it = pcg_solve(hv, make_prec(use), b, max_iter, tol_rate, d0, info)
    if info["neg_curv"] and state is not None and model is not None:
        state["psd_safe"] = True  # (sticky: reported in the step's flags)
        model.psd_safe = True
        d, it2 = pcg_solve(hv, make_prec(use), b, max_iter, tol_rate, d0, info)
        it += it2
    if info["warm_discard"]:
        d, it2 = pcg_solve(hv, make_prec(use), b, max_iter, tol_rate, None, info)
        it += it2
    if use is not None and info["res_ratio"] is not None and info["res_ratio"] > COARSE_TRUST**2:
        if state is not None:
            state["coarse_off"] = True
        d, it2 = pcg_solve(hv, make_prec(None), b, max_iter, tol_rate, d0)
        it += it2
    return d, it


def make_preconditioner(Dinv, coarse=None):
    """z = D^-1 r (3x3 block Jacobi) [+ P A_c^-1 P^T r: additive coarse-grid correction, `coarse` = (node (V,8), weight (V,8),
    inverse coarse operator), the tables tacex_fem_set_coarse_space takes]."""
    if coarse is None:
        return lambda r: np.einsum("vij,vj->vi", Dinv, r)
    node, w, aci = coarse
    nc = aci.shape[0] // 3
    V = node.shape[0]

    def prec(r):
        rc = np.zeros((nc, 3))
        np.add.at(rc, node.reshape(-1), (w[:, :, None] * r[:, None, :]).reshape(-1, 3))
        yc = (aci @ rc.reshape(-1)).reshape(nc, 3)
        return np.einsum("vij,vj->vi", Dinv, r) + (w[:, :, None] * yc[node]).sum(1)

    return prec


def chain_tables(chains, V):
    """chains: list of vertex-id sequences (each vertex in at most one; the rest are singletons) -> (next (V,) with -1 at a chain's
    end, heads: first vertex of every chain, singletons included, ascending by head)."""
    nxt = np.full(V, -1, np.int64)
    is_member = np.zeros(V, bool)
    heads = []
    for ch in (chains or []):
        ch = [int(v) for v in ch]
        if not ch:
            continue
        assert not is_member[ch].any(), "a vertex belongs to two chains"
        is_member[ch] = True
        heads.append(ch[0])
        for a, b in zip(ch[:-1], ch[1:]):
            nxt[a] = b
    heads += [v for v in range(V) if not is_member[v]]
    return nxt, np.array(sorted(heads), np.int64)


def _inv3_spd(A):
    """The kernel's inv3_spd: Cholesky pivots must be positive; None otherwise."""
    try:
        np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        return None
    return np.linalg.inv(A)


def chain_factor(D, E, nxt, heads):
    """Block-tridiagonal LDL^T along vertex chains - the block part of the kernel's preconditioner.  D (V,3,3): diagonal blocks of the
    system matrix, E (V,3,3): its block (v, next(v)) (zero where v ends a chain).  Walks every chain from its head:
        S_0 = D_0,  S_i = D_i - E_{i-1}^T G_{i-1},  G_i = S_i^-1 E_i
    and returns (S^-1, G) ROUNDED TO float32 - the kernel keeps them in LDS as floats (6 + 9 per vertex).  Where S_i is not positive
    definite (cannot happen in exact arithmetic with PSD-projected element Hessians) the block falls back to I / max diag S_i.  A chain
    of one vertex is plain block Jacobi.  The operator L^-T S^-1 L^-1 is symmetric positive definite for ANY G as long as the S^-1
    are, so rounding G costs quality, not validity."""
    V = D.shape[0]
    Sinv = np.zeros((V, 3, 3))
    G = np.zeros((V, 3, 3))
    for h in heads:
        v, S = int(h), D[int(h)].copy()
        while True:
            Si = _inv3_spd(S)
            if Si is None:
                dmax = max(S[0, 0], S[1, 1], S[2, 2])
                Si = np.eye(3) / (dmax if dmax > 0 else 1.0)
            Sinv[v] = Si
            n = int(nxt[v])
            if n < 0:
                break
            G[v] = Si @ E[v]
            S = D[n] - E[v].T @ G[v]
            v = n
    # symmetric part of S^-1 is what the 6 stored floats hold (upper triangle)
    Sinv = np.triu(Sinv) + np.swapaxes(np.triu(Sinv, 1), -1, -2)
    return Sinv.astype(np.float32).astype(np.float64), G.astype(np.float32).astype(np.float64)


def make_chain_preconditioner(Sinv, G, nxt, heads, coarse=None):
    """z = L^-T S^-1 L^-1 r along the chains (forward: y_i = r_i - G_{i-1}^T y_{i-1}; t_i = S_i^-1 y_i; backward: z_i = t_i - G_i z_{i+1})
    [+ the additive coarse-grid correction of make_preconditioner]."""
    V = Sinv.shape[0]
    chains = []
    for h in heads:
        ch, v = [], int(h)
        while v >= 0:
            ch.append(v)
            v = int(nxt[v])
        chains.append(ch)
    cpart = make_preconditioner(np.zeros((V, 3, 3)), coarse) if coarse is not None else None

    def prec(r):
        z = np.zeros_like(r)
        for ch in chains:
            y = np.zeros((len(ch), 3))
            for i, v in enumerate(ch):
                y[i] = r[v] - (G[ch[i - 1]].T @ y[i - 1] if i > 0 else 0.0)
            for i in range(len(ch) - 1, -1, -1):
                v = ch[i]
                z[v] = Sinv[v] @ y[i] - (G[v] @ z[ch[i + 1]] if i + 1 < len(ch) else 0.0)
        return z + cpart(r) if cpart is not None else z

    return prec


def lame_from_youngs_poisson(E: float, nu: float):
    return E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))


@dataclass
class FemModel:
    X: np.ndarray         # (V,3) rest positions
    tets: np.ndarray      # (T,4), positively oriented
    dm_inv: np.ndarray    # (T,3,3)
    vol: np.ndarray       # (T,)
    mass: np.ndarray      # (V,) lumped
    mu: float
    lam: float
    alpha: float
    psi_rest: float
    dt: float
    strength: float

    @classmethod
    def build(cls, X, tets, youngs=1e4, poisson=0.49, density=1e3, dt=0.01, strength=100.0):
        X = np.asarray(X, np.float64)
        tets = np.array(tets, dtype=np.int32)
        Dm = np.stack([X[tets[:, 1]] - X[tets[:, 0]], X[tets[:, 2]] - X[tets[:, 0]], X[tets[:, 3]] - X[tets[:, 0]]], -1)
        det = np.linalg.det(Dm)
        flip = det < 0
        tets[flip] = tets[flip][:, [0, 2, 1, 3]]  # re-orient
        Dm = np.stack([X[tets[:, 1]] - X[tets[:, 0]], X[tets[:, 2]] - X[tets[:, 0]], X[tets[:, 3]] - X[tets[:, 0]]], -1)
        vol = np.linalg.det(Dm) / 6.0
        assert (vol > 0).all(), "degenerate tetrahedron"
        mass = np.zeros(len(X))
        np.add.at(mass, tets.reshape(-1), np.repeat(density * vol / 4.0, 4))
        mu_l, lam_l = lame_from_youngs_poisson(youngs, poisson)
        mu, lam = 4.0 / 3.0 * mu_l, lam_l + 5.0 / 6.0 * mu_l
        alpha = 1.0 + 0.75 * mu / lam
        psi_rest = 0.5 * lam * (1 - alpha) ** 2 - 0.5 * mu * np.log(4.0)
        return cls(X, tets, np.linalg.inv(Dm), vol, mass, mu, lam, alpha, psi_rest, dt, strength)

    # ---- kinematics --------------------------------------------------------------------------------
    def deformation_gradient(self, x):
        """x (..., V, 3) -> F (..., T, 3, 3)."""
        t = self.tets
        Ds = np.stack([x[..., t[:, 1], :] - x[..., t[:, 0], :], x[..., t[:, 2], :] - x[..., t[:, 0], :],
                       x[..., t[:, 3], :] - x[..., t[:, 0], :]], -1)
        return Ds @ self.dm_inv

    @staticmethod
    def cofactor(F):
        f0, f1, f2 = F[..., :, 0], F[..., :, 1], F[..., :, 2]
        return np.stack([np.cross(f1, f2), np.cross(f2, f0), np.cross(f0, f1)], -1)

    # ---- element terms -----------------------------------------------------------------------------------
    def psi(self, F):
        Ic = (F * F).sum((-2, -1))
        J = np.linalg.det(F)
        return 0.5 * self.mu * (Ic - 3) + 0.5 * self.lam * (J - self.alpha) ** 2 - 0.5 * self.mu * np.log(Ic + 1) - self.psi_rest

    def pk1(self, F):
        Ic = (F * F).sum((-2, -1))[..., None, None]
        J = np.linalg.det(F)[..., None, None]
        return self.mu * (1 - 1 / (Ic + 1)) * F + self.lam * (J - self.alpha) * self.cofactor(F)

    def dpk1(self, F, dF):
        """Directional derivative of P (the 9x9 Hessian applied to dF)."""
        Ic = (F * F).sum((-2, -1))[..., None, None]
        J = np.linalg.det(F)[..., None, None]
        C = self.cofactor(F)
        f0, f1, f2 = F[..., :, 0], F[..., :, 1], F[..., :, 2]
        d0, d1, d2 = dF[..., :, 0], dF[..., :, 1], dF[..., :, 2]
        dC = np.stack([np.cross(d1, f2) + np.cross(f1, d2), np.cross(d2, f0) + np.cross(f2, d0),
                       np.cross(d0, f1) + np.cross(f0, d1)], -1)
        FdF = (F * dF).sum((-2, -1))[..., None, None]
        CdF = (C * dF).sum((-2, -1))[..., None, None]
        a = self.mu * (1 - 1 / (Ic + 1))
        c = self.lam * (J - self.alpha)
        if getattr(self, "psd_safe", False):
            # PSD-SAFE MODE (an env switches to it for the rest of a time step once its PCG has met negative curvature): the only
            # indefinite term of the Hessian is c d2J/dF2, whose spectral norm is <= sqrt(2 Ic) (eigenvalues +-sigma_k and those of the
            # scaling block, bounded by two singular values); with |c| clamped to a / sqrt(2 Ic) the sum a I + c d2J/dF2 stays positive
            # semi-definite and so does the element Hessian (the other two terms are rank-one PSD).  The gradient is untouched: Newton
            # becomes a quasi-Newton iteration on the same minimiser.
            lim = a / np.sqrt(2.0 * np.maximum(Ic, 1e-300))
            c = np.clip(c, -lim, lim)
        return a * dF + 2 * self.mu / (Ic + 1) ** 2 * FdF * F + self.lam * CdF * C + c * dC

    def element_energy(self, x):
        return self.vol * self.psi(self.deformation_gradient(x))  # (..., T)

    def element_gradient(self, x):
        """(..., T, 12): gradient of vol*Psi wrt the 4 vertices (x0,x1,x2,x3), xyz fastest."""
        F = self.deformation_gradient(x)
        Hm = self.vol[:, None, None] * (self.pk1(F) @ np.swapaxes(self.dm_inv, -1, -2))  # columns: d/dx1..3
        g1, g2, g3 = Hm[..., :, 0], Hm[..., :, 1], Hm[..., :, 2]
        return np.concatenate([-(g1 + g2 + g3), g1, g2, g3], -1)

    def element_hessian(self, x, project_psd=False):
        """(..., T, 12, 12) Hessian of vol*Psi; optional projection of the 9x9 F-space Hessian onto PSD."""
        F = self.deformation_gradient(x)
        shp = F.shape[:-2]
        # G: (T, 9, 12) with vec(F) row-major (i*3+m) = G @ x_e
        T = len(self.tets)
        G = np.zeros((T, 9, 12))
        r = np.concatenate([-self.dm_inv.sum(-2, keepdims=True), self.dm_inv], -2)  # (T,4,3): dF[k,m]/dx[v,k] = r[v,m]
        for v in range(4):
            for k in range(3):
                for m in range(3):
                    G[:, k * 3 + m, v * 3 + k] = r[:, v, m]
        H9 = np.zeros(shp + (9, 9))
        safe, self.psd_safe = getattr(self, "psd_safe", False), False  # assembled blocks (preconditioner, coarse operator) are always the exact ones
        for q in range(9):
            dF = np.zeros(shp + (3, 3))
            dF[..., q // 3, q % 3] = 1.0
            H9[..., :, q] = self.dpk1(F, dF).reshape(shp + (9,))
        self.psd_safe = safe
        if project_psd:
            w, V = np.linalg.eigh(0.5 * (H9 + np.swapaxes(H9, -1, -2)))
            H9 = (V * np.maximum(w, 0.0)[..., None, :]) @ np.swapaxes(V, -1, -2)
        return self.vol[:, None, None] * (np.swapaxes(G, -1, -2) @ H9 @ G)

    # ---- incremental potential ------------------------------------------------------------------------------
    def energy(self, x, x_tilde, constrained=None, aim=None):
        d = x - x_tilde
        E = 0.5 * (self.mass[:, None] * d * d).sum((-2, -1)) + self.dt**2 * self.element_energy(x).sum(-1)
        if constrained is not None:
            c = x - aim
            E = E + 0.5 * self.strength * (constrained[..., None] * self.mass[:, None] * c * c).sum((-2, -1))
        return E

    def gradient(self, x, x_tilde, constrained=None, aim=None):
        g = self.mass[:, None] * (x - x_tilde)
        ge = self.element_gradient(x).reshape(x.shape[:-2] + (len(self.tets), 4, 3)) * self.dt**2
        for b in np.ndindex(x.shape[:-2]):
            np.add.at(g[b], self.tets.reshape(-1), ge[b].reshape(-1, 3))
        if constrained is not None:
            g = g + self.strength * constrained[..., None] * self.mass[:, None] * (x - aim)
        return g

    def hess_vec(self, x, p, constrained=None):
        """(M + dt^2 K + s M_c) p, matrix-free through dpk1."""
        F = self.deformation_gradient(x)
        dF = self.deformation_gradient(p)  # linear in p (rest offsets cancel: Ds is a difference)
        dH = self.vol[:, None, None] * (self.dpk1(F, dF) @ np.swapaxes(self.dm_inv, -1, -2)) * self.dt**2
        g1, g2, g3 = dH[..., :, 0], dH[..., :, 1], dH[..., :, 2]
        ge = np.stack([-(g1 + g2 + g3), g1, g2, g3], -2)  # (..., T, 4, 3)
        y = self.mass[:, None] * p
        for b in np.ndindex(x.shape[:-2]):
            np.add.at(y[b], self.tets.reshape(-1), ge[b].reshape(-1, 3))
        if constrained is not None:
            y = y + self.strength * constrained[..., None] * self.mass[:, None] * p
        return y

    def diag_blocks(self, x, constrained=None):
        """(..., V, 3, 3) diagonal blocks of M + dt^2 K + s M_c (block-Jacobi preconditioner)."""
        He = self.element_hessian(x) * self.dt**2
        D = np.zeros(x.shape[:-2] + (len(self.X), 3, 3))
        for b in np.ndindex(x.shape[:-2]):
            for v in range(4):
                np.add.at(D[b], self.tets[:, v], He[b][:, v * 3 : v * 3 + 3, v * 3 : v * 3 + 3])
        m = self.mass * (1 if constrained is None else 1)
        D = D + m[:, None, None] * np.eye(3)
        if constrained is not None:
            D = D + (self.strength * constrained * self.mass)[..., None, None] * np.eye(3)
        return D

    def offdiag_blocks(self, x, nxt):
        """(V,3,3): block (v, nxt[v]) of dt^2 K (zero where nxt[v] < 0 or the two share no tet) - the coupling the chain
        preconditioner keeps (single env)."""
        He = self.element_hessian(x) * self.dt**2
        E = np.zeros((len(self.X), 3, 3))
        for a in range(4):
            for b in range(4):
                if a == b:
                    continue
                sel = nxt[self.tets[:, a]] == self.tets[:, b]
                if sel.any():
                    np.add.at(E, self.tets[sel, a], He[sel][:, a * 3 : a * 3 + 3, b * 3 : b * 3 + 3])
        return E

    def block_preconditioner(self, x, D, mdiag, coarse=None, chains=None):
        """The preconditioner of the Newton kernels from the diagonal blocks D: chain factor (chains = (next, heads) of
        chain_tables; None: every vertex its own chain = block Jacobi, inverse blocks stored as float32) + coarse correction."""
        V = len(self.X)
        nxt, heads = chains if chains is not None else (np.full(V, -1, np.int64), np.arange(V))
        E = self.offdiag_blocks(x, nxt) if (nxt >= 0).any() else np.zeros((V, 3, 3))
        Sinv, G = chain_factor(D, E, nxt, heads)
        return make_chain_preconditioner(Sinv, G, nxt, heads, coarse)

    # ---- one Newton iteration: truncated PCG + backtracking line search (US:70-76) ----------------------------------
    def newton_step(self, x, x_tilde, constrained=None, aim=None, pcg_max_iter=64, pcg_tol_rate=1e-3, ls_max_iter=8, coarse=None, d0=None,
                    return_dir=False, chains=None, state=None):
        """Single env (x: (V,3)).  Returns (x_new, stats=[E0, E1, step, pcg_iters, max |d|, ccd step]).  `coarse` = (node (V,8),
        weight (V,8), inverse coarse operator (3 nc, 3 nc)): the additive coarse-grid correction of tacex_fem_set_coarse_space.
        `state`: see pcg_solve_guarded."""
        g = self.gradient(x, x_tilde, constrained, aim)
        D = self.diag_blocks(x, constrained)
        mdiag = self.mass * (1.0 + (self.strength * constrained if constrained is not None else 0.0))
        d, it = pcg_solve_guarded(lambda p: self.hess_vec(x, p, constrained), lambda cs: self.block_preconditioner(x, D, mdiag, cs, chains), -g,
                                  pcg_max_iter, pcg_tol_rate, d0, coarse, state, self)
        E0 = self.energy(x, x_tilde, constrained, aim)
        step = direction_cap(self, d)
        E1 = E0
        x_new = x
        for _ in range(max(ls_max_iter, LS_RESCUE) + 1):  # capped search, then the rescue halvings (see LS_RESCUE)
            cand = x + step * d
            Ec = self.energy(cand, x_tilde, constrained, aim)
            if Ec <= E0:
                x_new, E1 = cand, Ec
                break
            step *= 0.5
        else:
            step = 0.0
        st = np.array([E0, E1, step, it, np.abs(d).max(), 1.0])
        return (x_new, st, d) if return_dir else (x_new, st)


def marker_uv(surf_pos, tri, weight, fx=340.0, fy=325.0, cx=160.0, cy=125.0):
    """FEM-driven markers (tactile_sensor_sapienipc_modified.py:347-366): barycentric point on a surface triangle,
    then pinhole projection uv = (K p) / p_z.   surf_pos (B,Vs,3) camera frame, tri (M,3), weight (M,3) -> (B,M,2)."""
    pts = (surf_pos[:, tri, :] * weight[None, :, :, None]).sum(-2)
    u = fx * pts[..., 0] / pts[..., 2] + cx
    v = fy * pts[..., 1] / pts[..., 2] + cy
    return np.stack([u, v], -1)


def attachment_aim_positions(offsets, body_pos, body_quat):
    """Attachment targets (uipc_attachments.py:387-428): IsaacLab `transform_points(offsets, pos, quat)` = R(q) offsets + pos in
    float32, R = `matrix_from_quat` (wxyz).  offsets (A,3), body_pos (B,3), body_quat (B,4) -> (B,A,3) float64."""
    f = np.float32
    q = np.asarray(body_quat, f)
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    two_s = f(2.0) / (q * q).sum(-1, dtype=f)
    R = np.stack([f(1) - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                  two_s * (i * j + k * r), f(1) - two_s * (i * i + k * k), two_s * (j * k - i * r),
                  two_s * (i * k - j * r), two_s * (j * k + i * r), f(1) - two_s * (i * i + j * j)], -1).reshape(-1, 3, 3).astype(f)
    out = np.einsum("bij,aj->bai", R, np.asarray(offsets, f)).astype(f) + np.asarray(body_pos, f)[:, None, :]
    return out.astype(np.float64)


# --------------------------------------------------------------------------------------------------
# IPC contact of the surface vertices against one analytic indenter (first slice).  PARITY UNPINNED like the
# rest of this file (libuipc's contact lives in the absent submodule); restates Li et al. 2020, "Incremental Potential
# Contact", eq. 6:  b(d) = -(d - dhat)^2 ln(d / dhat) on 0 < d < dhat, here in the dimensionless gap s = d / dhat and weighted
# per vertex:  E_c(x) = dt^2 kappa sum_v area_v b(d_v / dhat).   Known-answer tests: tests/test_fem_oracle.py.
# --------------------------------------------------------------------------------------------------
def closest_point_on_triangles(p, a, b, c):
    """Closest points of triangles (a, b, c: (Nt,3)) to points p (Np,3) -> (Np,Nt,3): Ericson, Real-Time Collision Detection 5.1.5,
    region by region in the book's order (vertex a, vertex b, edge ab, vertex c, edge ac, edge bc, face) - the order the kernel's
    mesh_distance tests them in."""
    p = np.asarray(p, np.float64)[:, None, :]
    ab, ac = (b - a)[None], (c - a)[None]
    ap = p - a[None]
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b[None]
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c[None]
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        t_ab = d1 / (d1 - d3)
        t_ac = d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        den = 1.0 / (va + vb + vc)
    conds = [
        (d1 <= 0) & (d2 <= 0),
        (d3 >= 0) & (d4 <= d3),
        (vc <= 0) & (d1 >= 0) & (d3 <= 0),
        (d6 >= 0) & (d5 <= d6),
        (vb <= 0) & (d2 >= 0) & (d6 <= 0),
        (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0),
    ]
    A, B, C = a[None], b[None], c[None]
    cands = [
        np.broadcast_to(A, ap.shape),
        np.broadcast_to(B, ap.shape),
        A + t_ab[..., None] * ab,
        np.broadcast_to(C, ap.shape),
        A + t_ac[..., None] * ac,
        B + t_bc[..., None] * (C - B),
    ]
    q = A + ab * (vb * den)[..., None] + ac * (vc * den)[..., None]
    for cond, cand in reversed(list(zip(conds, cands))):  # first matching region wins
        q = np.where(cond[..., None], cand, q)
    return q


def rotation_from_vector(r):
    """Rodrigues: rotation matrix of the rotation vector r (axis * angle) - the pose a mesh indenter row carries."""
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def contact_distance(ind, x, mesh=None):
    """Signed distance d (V,) and its gradient n (V,3) of points x (V,3) to indenter [kind, cx, cy, cz, R, nx, ny, nz]
    (kind 1 sphere, 2 half-space with unit normal n, 3 capsule with half axis vector n and radius R, 4 rigid triangle mesh `mesh` =
    (vertices (Nv,3) in its own frame, triangles (Nt,3)) placed at c with rotation vector n and inflated by R: UNSIGNED distance to
    the nearest triangle minus R, first triangle on ties)."""
    kind = int(ind[0])
    c = np.asarray(ind[1:4], np.float64)
    if kind == 1:
        r = x - c
        rho = np.linalg.norm(r, axis=-1)
        return rho - ind[4], r / np.maximum(rho, 1e-300)[..., None]
    if kind == 2:
        n = np.asarray(ind[5:8], np.float64)
        return (x - c) @ n, np.broadcast_to(n, x.shape).copy()
    if kind == 3:  # capsule: (nx, ny, nz) is HALF the axis vector, R the radius
        a = np.asarray(ind[5:8], np.float64)
        p = x - c
        aa = a @ a
        t = np.clip((p @ a) / aa, -1.0, 1.0) if aa > 0 else np.zeros(x.shape[:-1])
        r = p - t[..., None] * a
        rho = np.linalg.norm(r, axis=-1)
        return rho - ind[4], r / np.maximum(rho, 1e-300)[..., None]
    if kind == 4:
        assert mesh is not None, "indenter kind 4 needs the mesh"
        vm, tm = np.asarray(mesh[0], np.float64), np.asarray(mesh[1], np.int64)
        Rm = rotation_from_vector(ind[5:8])
        pl = (x - c) @ Rm  # into the mesh frame: R^T (x - c)
        q = closest_point_on_triangles(pl, vm[tm[:, 0]], vm[tm[:, 1]], vm[tm[:, 2]])
        r = pl[:, None, :] - q
        d2 = (r * r).sum(-1)
        k = np.argmin(d2, axis=1)  # first minimum
        rk = r[np.arange(len(pl)), k]
        rho = np.sqrt(d2[np.arange(len(pl)), k])
        return rho - ind[4], (rk / np.maximum(rho, 1e-300)[..., None]) @ Rm.T
    return np.full(x.shape[:-1], np.inf), np.zeros_like(x)


def barrier(s):
    """b(s), b'(s), b''(s) of the dimensionless IPC barrier, 0 outside (0, 1); +inf energy for s <= 0."""
    s = np.asarray(s, np.float64)
    act = (s > 0) & (s < 1)
    ss = np.where(act, s, 0.5)
    ln, q = np.log(ss), ss - 1.0
    b = np.where(act, -q * q * ln, 0.0)
    b = np.where(s <= 0, np.inf, b)
    b1 = np.where(act, -2 * q * ln - q * q / ss, 0.0)
    b2 = np.where(act, -2 * ln - 4 * q / ss + q * q / (ss * ss), 0.0)
    return b, b1, b2


class ContactModel:
    """Barrier terms of one env: `area` (V,) vertex weights, indenter row, dhat [m], kappa [J/m^2], dt."""

    def __init__(self, area, indenter, dhat, kappa, dt, mesh=None):
        self.area, self.ind, self.dhat, self.kappa, self.dt = np.asarray(area, np.float64), np.asarray(indenter, np.float64), dhat, kappa, dt
        self.mesh = mesh  # (vertices, triangles) of a kind-4 indenter

    def energy(self, x):
        d, _ = contact_distance(self.ind, x, self.mesh)
        b, _, _ = barrier(d / self.dhat)
        with np.errstate(invalid="ignore"):
            e = np.where(self.area > 0, self.area * b, 0.0)
        return self.dt**2 * self.kappa * e.sum()

    def gradient(self, x):
        d, n = contact_distance(self.ind, x, self.mesh)
        _, b1, _ = barrier(d / self.dhat)
        return (self.dt**2 * self.kappa * self.area * b1 / self.dhat)[:, None] * n

    def hess_blocks(self, x):
        """(V,3,3) PSD-projected diagonal blocks b'' n n^T (the b' hess(d) part is dropped, as IPC does)."""
        d, n = contact_distance(self.ind, x, self.mesh)
        _, _, b2 = barrier(d / self.dhat)
        return (self.dt**2 * self.kappa * self.area * b2 / self.dhat**2)[:, None, None] * n[:, :, None] * n[:, None, :]

    def max_step(self, x, dx, slack=0.9):
        """CCD filter: the largest step in [0, 1] that keeps every weighted vertex at a positive gap (1-Lipschitz bound)."""
        d, _ = contact_distance(self.ind, x, self.mesh)
        nd = np.linalg.norm(dx, axis=-1)
        ok = (self.area > 0) & (nd > 0) & (d > 0) & np.isfinite(d)
        return float(min(1.0, (slack * d[ok] / nd[ok]).min())) if ok.any() else 1.0


def friction_f0(y, eps):
    """IPC's smoothed friction potential per unit of normal force and friction ratio (Li et al. 2020, eq. 18-20): f0(y) =
    -y^3/(3 eps^2) + y^2/eps + eps/3 below the stick tolerance eps, y beyond; returns (f0, f1 / y, f1') with f1 = f0'."""
    y = np.asarray(y, np.float64)
    st = y < eps
    f0 = np.where(st, -y**3 / (3 * eps * eps) + y * y / eps + eps / 3, y)
    f1_over_y = np.where(st, 2 / eps - y / (eps * eps), 1 / np.maximum(y, 1e-300))
    f1p = np.where(st, 2 / eps - 2 * y / (eps * eps), 0.0)
    return f0, f1_over_y, f1p


class FrictionModel:
    """Coulomb friction of the gelpad surface against its env's analytic indenter (uipc_sim.py:103-124: `enable_friction`,
    `default_friction_ratio`, `eps_velocity`), the IPC way (Li et al. 2020, eq. 18-20): normal force lam = -dB/dd and contact normal n
    are LAGGED - frozen at a given state (`update(x)`) - so the friction potential
        D(x) = dt^2 mu lam f0(|u|),   u = (I - n n^T) (x - x_n - disp)
    (x_n = positions the time step started from, disp = the indenter's own displacement since the previous step; u = tangential sliding
    relative to it) is a smooth function of x with gradient dt^2 mu lam (f1 / |u|) u and the positive semi-definite Hessian
    dt^2 mu lam [(f1 / |u|) (T - t t^T) + f1' t t^T], T = I - n n^T, t = u / |u| (both coefficients >= 0: no projection needed).
    WHICH state the lag is taken from: the one the step starts from (`fem_step(friction_lag="start")`, the kernel's default) - IPC's lag
    "from the previous time step".  After the indenter has moved that state sits deep in the 10 GPa barrier, where -dB/dd is orders of
    magnitude above the elastic forces (Newton directions of metres, PCG at its cap) - the reason rounds 3-4 solved the step in two
    phases (`friction_lag="converged"`: normal contact alone until converged, then the lag from that balanced state and a friction
    phase); `update` however takes the smaller of -dB/dd and the contact reaction, and at the start state - the previous step's
    equilibrium - that reaction IS the previous normal force.  eps = eps_velocity * dt."""

    def __init__(self, cm: ContactModel, x_n, disp, mu, eps_velocity):
        self.cm, self.dt, self.mu, self.eps = cm, cm.dt, float(mu), float(eps_velocity) * cm.dt
        self.x_n = np.asarray(x_n, np.float64)
        self.disp = np.asarray(disp, np.float64)
        self.update(self.x_n)

    def update(self, x, g_other=None):
        """Freeze normal force and normal at the iterate x (the start of a Newton iteration).  `g_other` (V,3): gradient of the
        step's potential WITHOUT the contact terms (inertia + elasticity + constraints) at x.  In force balance the barrier force on
        a vertex is exactly the reaction to it, lam = (g_other . n) / dt^2; the lag takes the SMALLER of the two.  Why: the Newton
        loop stops on its step-size tolerance (velocity_tol * dt = 0.5 mm, US:62-66), where a contact vertex may still sit at
        0.98 d_hat - there the 10 GPa barrier pushes with 87 N on a pad whose whole reaction is below 1 N, the lagged friction force
        is two orders of magnitude above anything the gel can oppose, the first friction iteration's direction measures 0.66 m and
        the env needs 34 Newton iterations / 1 900 PCG iterations (scene step 11, env 442: profiles/r04_experiments.md) while the
        launch waits for it.  At convergence both expressions agree, so the capped lag is IPC's lag wherever IPC's is meaningful."""
        cm = self.cm
        d, n = contact_distance(cm.ind, x, cm.mesh)
        _, b1, _ = barrier(d / cm.dhat)
        with np.errstate(invalid="ignore"):
            self.lam = np.where(cm.area > 0, -cm.kappa * cm.area * b1 / cm.dhat, 0.0)  # normal force [N] per vertex, >= 0
            if g_other is not None:
                react = np.maximum((np.asarray(g_other, np.float64) * n).sum(-1) / self.dt**2, 0.0)
                self.lam = np.where(self.lam > 0, np.minimum(self.lam, react), self.lam)
        self.n = np.where((self.lam > 0)[:, None], n, 0.0)

    def _u(self, x):
        r = x - self.x_n - self.disp
        u = r - (r * self.n).sum(-1, keepdims=True) * self.n
        return u, np.linalg.norm(u, axis=-1)

    def energy(self, x):
        _, y = self._u(x)
        return self.dt**2 * self.mu * (self.lam * friction_f0(y, self.eps)[0])[self.lam > 0].sum()

    def gradient(self, x):
        u, y = self._u(x)
        return (self.dt**2 * self.mu * self.lam * friction_f0(y, self.eps)[1])[:, None] * u

    def hess_blocks(self, x):
        u, y = self._u(x)
        _, a, bq = friction_f0(y, self.eps)
        t = u / np.maximum(y, 1e-300)[:, None]
        T = np.eye(3)[None] - self.n[:, :, None] * self.n[:, None, :]
        tt = t[:, :, None] * t[:, None, :]
        return (self.dt**2 * self.mu * self.lam)[:, None, None] * (a[:, None, None] * (T - tt) + bq[:, None, None] * tt)


def edge_snap(m, cm, x, x_tilde, constrained, aim, x_prec=None):
    """One exact 1-D minimisation per surface vertex that is about to run into the barrier zone from outside (or sits in its outermost
    sliver), along its contact normal - a nonlinear Gauss-Seidel sweep over the stiffest degrees of freedom, taken before the Newton
    system of an iteration is set up (fem_newton_lds_kernel, same rule).  The barrier is C2 with b'' -> 0 at d_hat: the Newton system
    is blind to it for such a vertex, its direction sends the vertex a millimetre deep into a wall that stops it within microns, the
    line search cuts the step OF THE WHOLE MESH to a per cent and the next iteration repeats it (the apex vertex of a retreating
    contact crossed the zone edge back and forth for 13 iterations; tests/studies/fem_straggler_replay.py).  Along n the vertex's
    energy is   phi(t) = -(g.n) t + 1/2 (n.D n) t^2 + dt^2 kappa A b((gap - t) / d_hat),   g = contact-free gradient, D = diagonal
    block: if the elastic 1-D Newton step t_el = g.n / n.D n reaches the zone, the vertex is moved to where the barrier balances
    the force it has to carry, e = sqrt(lam d_hat / (3 kappa A)) below d_hat (b' ~ -3 e^2 near the edge, lam = g.n / dt^2) - never
    further than t_el.  Energy decreases (a 1-D minimiser of a convex model); everything else is left to Newton."""
    g = m.gradient(x, x_tilde, constrained, aim)
    D = m.diag_blocks(x if x_prec is None else x_prec, constrained)
    gap, n = contact_distance(cm.ind, x, cm.mesh)
    with np.errstate(invalid="ignore", divide="ignore"):
        gn = (g * n).sum(-1)
        nDn = np.einsum("vi,vij,vj->v", n, D, n)
        t_el = np.where(nDn > 0, gn / nDn, 0.0)
        lam = np.maximum(gn, 0.0) / m.dt**2
        e = np.clip(np.sqrt(lam * cm.dhat / (3.0 * cm.kappa * np.where(cm.area > 0, cm.area, 1.0))), 1e-6, 1e-2)
        rest = gap - (1.0 - e) * cm.dhat          # distance to the balance depth (> 0: the vertex is shallower than that)
        snap = (cm.area > 0) & np.isfinite(gap) & (gn > 0) & (rest > 0) & (t_el > gap - cm.dhat)
        t = np.minimum(t_el, rest)
    if not snap.any():
        return x, 0
    x = x.copy()
    x[snap] -= t[snap, None] * n[snap]
    return x, int(snap.sum())


def newton_step_contact(m: "FemModel", cm: ContactModel, x, x_tilde, constrained=None, aim=None, pcg_max_iter=64, pcg_tol_rate=1e-3,
                        ls_max_iter=8, coarse=None, d0=None, return_dir=False, fr: "FrictionModel | None" = None, chains=None, x_prec=None, edge=True,
                        state=None):
    """`FemModel.newton_step` with the barrier terms of `cm` in gradient, preconditioner, H.p and energy, and the CCD step
    filter in front of the backtracking line search.  Returns (x_new, [E0, E1, step, pcg_iters]).  `x_prec`: the state the ELASTIC
    blocks of the preconditioner are taken at (tacex_fem_step lags them: assembled in the first Newton iteration of the step and
    reused by the later ones; barrier and friction blocks are always those of x)."""
    if edge:
        x, _ = edge_snap(m, cm, x, x_tilde, constrained, aim, x_prec)
    g = m.gradient(x, x_tilde, constrained, aim) + cm.gradient(x)
    Hc = cm.hess_blocks(x)
    if fr is not None:
        g = g + fr.gradient(x)
        Hc = Hc + fr.hess_blocks(x).astype(np.float32).astype(np.float64)  # the kernel keeps these blocks in LDS as floats
    xp = x if x_prec is None else x_prec
    D = m.diag_blocks(xp, constrained) + Hc
    mdiag = m.mass * (1.0 + (m.strength * constrained if constrained is not None else 0.0))
    hv = lambda p: m.hess_vec(x, p, constrained) + np.einsum("vij,vj->vi", Hc, p)
    energy = lambda y: m.energy(y, x_tilde, constrained, aim) + cm.energy(y) + (fr.energy(y) if fr is not None else 0.0)
    d, it = pcg_solve_guarded(hv, lambda cs: m.block_preconditioner(xp, D, mdiag, cs, chains), -g, pcg_max_iter, pcg_tol_rate, d0, coarse, state, m)
    E0 = energy(x)
    step = step0 = min(cm.max_step(x, d), direction_cap(m, d))
    E1, x_new = E0, x
    for _ in range(max(ls_max_iter, LS_RESCUE) + 1):
        cand = x + step * d
        Ec = energy(cand)
        if Ec <= E0:
            x_new, E1 = cand, Ec
            break
        step *= 0.5
    else:
        step = 0.0
    st = np.array([E0, E1, step, it, np.abs(d).max(), step0])
    return (x_new, st, d) if return_dir else (x_new, st)


def fem_step(m: "FemModel", cm, x, v, constrained=None, aim=None, gravity=(0.0, 0.0, -9.8), max_newton=8, velocity_tol=0.05,
             pcg_max_iter=1024, pcg_tol_rate=1e-3, ls_max_iter=8, coarse=None, friction=None, chains=None, indenter_disp=None, lag_prec=True,
             friction_lag="start"):
    """One backward-Euler step of ONE env the way `tacex_fem_step` runs it (what world.advance() does, US:250-252):
    x_tilde = x + dt v + dt^2 g; Newton iterations until the UNSCALED Newton direction of one has max |d| <= velocity_tol * dt
    (US:62-66; IPC's test on the search direction) or the cap; v = (x_new - x) / dt.  Returns (x_new, v_new, info) with
    info = [newton_iterations, max |d| of the last iteration, flags (2: a line search failed, 4: the coarse correction was dropped - COARSE_TRUST), pcg_iterations_total]."""
    x0 = x
    xt = x + m.dt * v + m.dt**2 * np.asarray(gravity, np.float64)
    # contact-following start (fem_newton_lds_kernel, `follow`): a surface vertex the indenter retreats from (disp . n < 0) and that
    # lands inside the barrier zone when moved by that normal component starts the Newton loop at x + (disp . n) n
    # (`indenter_disp`, default: the friction tuple's displacement).  An initial guess only.
    if indenter_disp is None and friction is not None:
        indenter_disp = friction[2]
    if cm is not None and indenter_disp is not None and np.any(np.asarray(indenter_disp) != 0.0):
        d0, n0 = contact_distance(cm.ind, x, cm.mesh)
        dn = n0 @ np.asarray(indenter_disp, np.float64)  # < 0: the indenter's surface moves away from the vertex (retreat)
        with np.errstate(invalid="ignore"):
            xm = x + dn[:, None] * n0
        cand = (cm.area > 0) & (dn < 0) & (d0 > 0) & np.isfinite(d0)
        dm, _ = contact_distance(cm.ind, np.where(cand[:, None], xm, x), cm.mesh)
        follow = cand & (dm > 0) & (dm < cm.dhat)
        x = np.where(follow[:, None], xm, x)
    # friction = (mu, eps_velocity, indenter displacement since the previous step), see FrictionModel.  friction_lag = "ipc": IPC's
    # previous-configuration lag (`tacex_fem_set_friction_lag(ctx, 1)`, below); "start" (the kernel's default, mode 0): normal force and normal are lagged at the state the step starts from - the previous step's equilibrium, IPC's
    # lag "from the previous time step" - and friction acts from the first iteration on; "converged" (rounds 3-4, TACEX_FEM_FRIC_LAG=0):
    # the lag is taken where this step's normal-contact solve converged, in a second phase of the loop
    fric_pending = friction is not None and cm is not None and friction_lag not in ("start", "ipc")
    fr = None
    if friction is not None and cm is not None and friction_lag == "ipc":
        # IPC's lag to the letter (Li et al. 2020, section 5.4): barrier force and normal of the PREVIOUS configuration - the start
        # positions against the indenter where it stood at the previous step; no cap, nothing of the current iterate
        ind_prev = np.array(cm.ind, np.float64)
        ind_prev[1:4] -= np.asarray(friction[2], np.float64)
        fr = FrictionModel(ContactModel(cm.area, ind_prev, cm.dhat, cm.kappa, cm.dt, cm.mesh), x0, friction[2], friction[0], friction[1])
        if not fr.lam.max() > 0.0:
            fr = None
    if friction is not None and cm is not None and friction_lag == "start":
        fr = FrictionModel(cm, x0, friction[2], friction[0], friction[1])
        fr.update(x, m.gradient(x, xt, constrained, aim))
        if not fr.lam.max() > 0.0:
            fr = None  # no vertex carries a normal force: this step runs without friction
    n, flags, pcg, dmax = 0, 0, 0, np.inf
    d0 = None
    # the elastic preconditioner blocks of the whole step are those of its first iteration (see newton_step_contact); lag_prec=False:
    # fresh blocks every iteration (the streaming kernel of meshes with more than 512 vertices)
    x_prec = x if lag_prec else None
    guard = {}  # state of the solver safeguards (pcg_solve_guarded): once fired, they stay on for the rest of this step
    m.psd_safe = False
    for _ in range(max_newton):
        if cm is not None:
            x, st, d = newton_step_contact(m, cm, x, xt, constrained, aim, pcg_max_iter, pcg_tol_rate, ls_max_iter, coarse, d0, True, fr, chains,
                                           x_prec, state=guard)
        else:
            x, st, d = m.newton_step(x, xt, constrained, aim, pcg_max_iter, pcg_tol_rate, ls_max_iter, coarse, d0, True, chains, state=guard)
        d0_used = d0
        d0 = (1.0 - st[2]) * d if 0.0 < st[2] < 1.0 else None  # warm start of the next PCG: the part of d that was cut off
        n += 1
        pcg += int(st[3])
        dmax = st[4]
        if st[2] == 0.0 and not dmax <= velocity_tol * m.dt:
            if d0_used is not None:
                continue  # the direction came from a warm start (no descent guarantee): once more from a zero start (d0 is None now)
            flags |= 2
            break  # a rejected search leaves x unchanged: every further iteration would repeat this one
        if dmax <= velocity_tol * m.dt:  # IPC's test: the unscaled search direction, whatever the CCD bound / line search made of the step
            if fric_pending:  # normal contact is balanced: take the friction lag from here and go on
                fric_pending = False
                fr = FrictionModel(cm, x0, friction[2], friction[0], friction[1])
                fr.update(x, m.gradient(x, xt, constrained, aim))
                if fr.lam.max() > 0.0:
                    continue
            break
    m.psd_safe = False
    if guard.get("coarse_off"):
        flags |= 4
    if guard.get("psd_safe"):
        flags |= 8
    return x, (x - x0) / m.dt, np.array([n, dmax, flags, pcg], np.float64)
