"""Replay on the CPU (oracle operators, dense) of the first Newton system of the dumped steps of scripts/r06/ball_dump.py: PCG iterations to the
reference's threshold (1e-3 on r.z) under the preconditioner variants of scripts/r06/coupled_coarse_probe.py, and the spectrum of M^-1 H."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.abd_oracle import AffineBody, BallScene
from oracle.fem_oracle import FemModel, pcg_solve
from tacex_amd.uipc.coarse_space import build_coarse_space, coarse_grid_dims, prolongation_matrix
from test_abd_oracle import _surface

Z_ = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_ball_states.npz")
P, T, vb, tb, back = Z_["P"], Z_["T"], Z_["vb"], Z_["tb"], Z_["back"]
dt, dhat, gh = float(Z_["dt"]), float(Z_["d_hat"]), float(Z_["gh"])
m = FemModel.build(P, T, dt=dt, strength=1000.0)
tri, area = _surface(m.X, m.tets)
sc = BallScene(m, tri, area, AffineBody(vb, tb, density=1e3), dhat=dhat, ground_height=gh)
sc.mu, sc.eps_v = 0.5, 0.01
V = sc.V
n = 3 * (V + 4)
cons = np.zeros(V); cons[back] = 1.0
node, w, nc = build_coarse_space(m.X, coarse_grid_dims(m.X))
Pm = prolongation_matrix(node, w, nc)


def dense(op, rows):
    k = 3 * rows
    H = np.zeros((k, k)); e = np.zeros((rows, 3))
    for j in range(k):
        e.reshape(-1)[j] = 1.0; H[:, j] = op(e).reshape(-1); e.reshape(-1)[j] = 0.0
    return H


A0 = dense(lambda p: m.hess_vec(m.X, p, cons), V)
A0c_inv = np.linalg.inv(Pm.T @ A0 @ Pm)
for k in range(4):
    if f"s{k}_x" not in Z_:
        break
    i, b, r = Z_[f"s{k}_meta"]
    y = np.concatenate([Z_[f"s{k}_x"], Z_[f"s{k}_q"]]); v = np.concatenate([Z_[f"s{k}_v"], Z_[f"s{k}_qv"]])
    aim = m.X.copy(); aim[back] = Z_[f"s{k}_aim"]
    yt = y + dt * v; g3 = dt * dt * np.array([0, 0, -9.8]); yt[:V] += g3; yt[V] += g3
    sc._lag = sc.friction_lag(y)
    pr = sc.pairs(y)
    print(f"== sample {k}: step {int(i)} env {int(b)}, GPU: newton {int(Z_[f's{k}_info'][0])} pcg {int(Z_[f's{k}_info'][3])}; pairs {[len(p[0]) for p in pr]} "
          f"d/d_hat {[np.round(p[3] / dhat, 3).tolist() for p in pr]} friction contacts {len(sc._lag[3])} lam {np.array2string(sc._lag[3], precision=3)}")
    H = dense(lambda p: sc.hess_vec(y, p, cons), V + 4); H = 0.5 * (H + H.T)
    g = sc.gradient(y, yt, cons, aim).reshape(-1)
    D = sc.diag_blocks(y, cons)[:V]; Dinv = np.linalg.inv(D); Bfull = H[3 * V:, 3 * V:]
    Zm = np.zeros((n, 3 * nc + 12)); Zm[:3 * V, :3 * nc] = Pm; Zm[3 * V:, 3 * nc:] = np.eye(12)
    Hc = H.copy(); Hc[:3 * V, :3 * V] -= dense(lambda p: m.hess_vec(y[:V], p, cons), V)
    Ac = Zm.T @ Hc @ Zm; Ac[:3 * nc, :3 * nc] += Pm.T @ A0 @ Pm
    def jac(r):
        z = np.zeros_like(r); z[:3 * V] = np.einsum("vij,vj->vi", Dinv, r[:3 * V].reshape(V, 3)).reshape(-1); return z
    def Ma(r):
        z = jac(r); z[3 * V:] = np.linalg.solve(Bfull, r[3 * V:]); return z
    def Mb(r):
        z = Ma(r); z[:3 * V] += Pm @ (A0c_inv @ (Pm.T @ r[:3 * V])); return z
    Aci = np.linalg.inv(Ac)
    def Mc(r):
        return jac(r) + Zm @ (Aci @ (Zm.T @ r))
    for name, M in (("a block Jacobi + ball block", Ma), ("b a + rest coarse (kernel, no chains)", Mb), ("c coupled coarse", Mc)):
        d, it = pcg_solve(lambda p: H @ p, M, -g, 4000, 1e-3)
        Mm = np.column_stack([M(e) for e in np.eye(n)])
        ev = np.sort(np.linalg.eigvals(Mm @ H).real)
        print(f"   {name:40s}: {it:4d} PCG iterations; eig(M^-1 H): min {ev[0]:.2e} next {ev[1]:.2e} {ev[2]:.2e} ... max {ev[-1]:.2f}; below 0.05: {(ev < 0.05).sum()}")

# ---- coarse grids of other sizes on the last sample (same H, g) ----
print("coarse grid variants on the last sample (block Jacobi + ball block + additive rest-state coarse correction):")
for dims in (coarse_grid_dims(m.X), (3, 4, 1), (3, 4, 2), (4, 5, 1), (4, 5, 2), (8, 10, 1)):
    nd, ww, ncc = build_coarse_space(m.X, dims)
    Pv = prolongation_matrix(nd, ww, ncc)
    Ai = np.linalg.inv(Pv.T @ A0 @ Pv)
    def Mv(r):
        z = Ma(r); z[:3 * V] += Pv @ (Ai @ (Pv.T @ r[:3 * V])); return z
    d, it = pcg_solve(lambda p: H @ p, Mv, -g, 4000, 1e-3)
    print(f"   dims {dims} ({ncc} nodes): {it} iterations")

# ---- PCG warm starts on the last sample (variant b = the kernel's preconditioner without chains); threshold stays 1e-3 of r.z of the ZERO guess ----
def pcg_warm(A, M, b, d0, tol, max_iter=4000):
    r = b.copy(); z = M(r); rz0 = r @ z
    d = d0.copy(); r = b - A(d); z = M(r); rz = r @ z; p = z.copy(); it = 0
    while it < max_iter and rz > tol * rz0:
        Ap = A(p); al = rz / (p @ Ap); d += al * p; r -= al * Ap; z = M(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return d, it
shift = (aim[back] - y[back]).mean(0)
d_rigid = np.zeros((V + 4, 3)); d_rigid[:V] = shift
d_pred = (yt - y)
d_pred_pad = d_pred.copy(); d_pred_pad[V:] = 0.0
print("mean constraint offset", shift, " max |x~ - x| pad", np.abs(d_pred[:V]).max(), " |solution| max", np.abs(np.linalg.solve(H, -g)).max())
for name, d0 in (("zero", np.zeros(n)), ("pad follows its constraints rigidly", d_rigid.reshape(-1)), ("x~ - x (all rows)", d_pred.reshape(-1)), ("x~ - x (pad rows)", d_pred_pad.reshape(-1))):
    d, it = pcg_warm(lambda p: H @ p, Mb, -g, d0, 1e-3)
    print(f"   warm start {name:40s}: {it} iterations")

# ---- line preconditioners on the last sample: exact inverses of H restricted to the vertex lines of the structured pad along z (what the kernel's
#      chains are), and additionally along x and y; always + ball block + the nested (4, 5, 1) coarse correction ----
from oracle.fem_oracle import box_tet_mesh  # noqa
Xr = m.X
def lines_along(axis):
    other = [a for a in range(3) if a != axis]
    key = np.round(Xr[:, other] * 1e7).astype(np.int64)
    groups = {}
    for v in range(V):
        groups.setdefault(tuple(key[v]), []).append(v)
    return [sorted(g, key=lambda v: Xr[v, axis]) for g in groups.values()]
def line_inverse(lines):
    blocks = []
    for ln in lines:
        idx = np.concatenate([[3 * v, 3 * v + 1, 3 * v + 2] for v in ln])
        blocks.append((idx, np.linalg.inv(H[np.ix_(idx, idx)])))
    def app(r):
        z = np.zeros_like(r)
        for idx, Bi_ in blocks:
            z[idx] = Bi_ @ r[idx]
        return z
    return app
nd, ww, ncc = build_coarse_space(m.X, (4, 5, 1))
Pv = prolongation_matrix(nd, ww, ncc)
Ai = np.linalg.inv(Pv.T @ A0 @ Pv)
Lz, Lx, Ly = (line_inverse(lines_along(a)) for a in (2, 0, 1))
print("lines:", [len(lines_along(a)) for a in (2, 0, 1)], "of lengths", [len(lines_along(a)[0]) for a in (2, 0, 1)])
def mk(parts, scale=1.0):
    def M(r):
        z = np.zeros_like(r)
        for p_ in parts:
            z += scale * p_(r)
        z[3 * V:] = np.linalg.solve(Bfull, r[3 * V:])
        z[:3 * V] += Pv @ (Ai @ (Pv.T @ r[:3 * V]))
        return z
    return M
for name, M in (("z lines (the kernel's chains) + coarse", mk([Lz])), ("z + x + y lines + coarse", mk([Lz, Lx, Ly])), ("(z + x + y) / 3 + coarse", mk([Lz, Lx, Ly], 1 / 3)),
                ("z + x lines + coarse", mk([Lz, Lx])), ("z + y lines + coarse", mk([Lz, Ly]))):
    for tol in (1e-3,):
        d, it = pcg_solve(lambda p: H @ p, M, -g, 4000, tol)
        print(f"   {name:45s}: {it} iterations")
