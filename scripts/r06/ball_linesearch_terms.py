"""Which term of the incremental potential rejects the full Newton step in the dumped light-contact states (scripts/r06/ball_dump.py)?  Oracle
operators on the CPU: the direction of the first Newton iteration, then every term of E along it."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.abd_oracle import AffineBody, BallScene
from oracle.fem_oracle import FemModel, barrier
from test_abd_oracle import _surface
Z_ = np.load("gpurun_out/r06_ball_states.npz")
P, T, vb, tb, back = Z_["P"], Z_["T"], Z_["vb"], Z_["tb"], Z_["back"]
dt, dhat, gh = float(Z_["dt"]), float(Z_["d_hat"]), float(Z_["gh"])
m = FemModel.build(P, T, dt=dt, strength=1000.0)
tri, area = _surface(m.X, m.tets)
sc = BallScene(m, tri, area, AffineBody(vb, tb, density=1e3), dhat=dhat, ground_height=gh)
sc.mu, sc.eps_v = 0.5, 0.01
V = sc.V
cons = np.zeros(V); cons[back] = 1.0
for k in range(4):
    y = np.concatenate([Z_[f"s{k}_x"], Z_[f"s{k}_q"]]); v = np.concatenate([Z_[f"s{k}_v"], Z_[f"s{k}_qv"]])
    aim = m.X.copy(); aim[back] = Z_[f"s{k}_aim"]
    yt = y + dt * v; g3 = dt * dt * np.array([0, 0, -9.8]); yt[:V] += g3; yt[V] += g3
    sc._lag = sc.friction_lag(y)
    y1, st = sc.newton_step(y, yt, cons, aim, pcg_max_iter=4000, pcg_tol_rate=1e-6)
    print(f"sample {k}: oracle newton step: E0 {st[0]:.6e} E1 {st[1]:.6e} step {st[2]:.4f} pcg {int(st[3])} dmax_x {st[4]:.3e} dmax_c {st[5]:.3e}")
    # direction again for term-wise energies along it
    from oracle.fem_oracle import pcg_solve
    g = sc.gradient(y, yt, cons, aim)
    d, it = pcg_solve(lambda p: sc.hess_vec(y, p, cons), sc.preconditioner(y, cons), -g, 4000, 1e-6)
    amax = sc.max_step(y, d)
    def terms(yy):
        x, q = yy[:V], yy[V:]
        dq = q - yt[V:]
        pad = sc.pad.energy(x, yt[:V], cons, aim)
        inert = 0.5 * np.einsum("ab,ai,bi->", sc.ball.S, dq, dq)
        ortho = dt**2 * sc.ball.ortho(q)[0]
        grd = sc._ground(x, sc.pad_area)[0] + sc._ground(sc.ball.points(q), sc.ball.area)[0]
        _, _, w, dd, _, mol, _ = sc._pair_rows(yy)
        prs = dt**2 * sc.kappa * (w * mol * barrier(dd / dhat)[0]).sum()
        return np.array([pad, inert, ortho, grd, prs]), len(dd)
    e0, n0 = terms(y)
    print(f"   amax {amax:.4f}; |d| affine rows max {np.abs(d[V+1:]).max():.3e}; pairs at start {n0}")
    for a in (amax, amax / 2, amax / 4, amax / 16, amax / 64):
        e, n = terms(y + a * d)
        print(f"   alpha {a:.5f}: dE total {(e - e0).sum(): .3e} | pad {e[0]-e0[0]: .2e} inertia {e[1]-e0[1]: .2e} ortho {e[2]-e0[2]: .2e} ground {e[3]-e0[3]: .2e} pairs {e[4]-e0[4]: .2e} (pairs {n})")
