"""CPU experiment (numpy, oracle operators): would a coarse space that COUPLES the pad's grid nodes with the ball's twelve unknowns cut the PCG
iterations of the ball scene?  The kernel's coarse correction uses the pad's rest-state Galerkin operator: it knows neither the ball nor the pairs.
Variants of M^-1 (block Jacobi on the pad everywhere; the chains are left out of all of them):
  a  block Jacobi + exact 12 x 12 ball block
  b  a + additive rest-state coarse correction on the pad (what fem_ball_newton_kernel does, minus the chains)
  c  block Jacobi + additive correction on Z = [P 0; 0 I_12] with A_c = rest-state pad operator + the pair / friction / ground terms restricted to Z
  d  as c with the exact Galerkin operator Z^T H Z (upper bound of what c can do)"""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_abd_oracle import _scene
from oracle.fem_oracle import pcg_solve
from tacex_amd.uipc.coarse_space import build_coarse_space, coarse_grid_dims, prolongation_matrix

press = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-4
sc, y, cons = _scene(press=press, level=2, R=0.009, mesh=(8, 10, 4), pole=True, shift=(0.0008, 0.0005))
sc.mu = 0.5
V = sc.V
n = 3 * (V + 4)
aim = y[:V].copy()
aim[:, 2] -= float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # the back face's target moves down by this much in the step
yt = y.copy(); yt[:V, 2] -= 9.8 * sc.dt**2; yt[V, 2] -= 9.8 * sc.dt**2
sc._lag = sc.friction_lag(y)
pr = sc.pairs(y)
print("pairs (PB, BP, EE):", [len(p[0]) for p in pr], "friction contacts:", len(sc._lag[3]))

def dense(op, shape_rows):
    m = 3 * shape_rows
    H = np.zeros((m, m))
    e = np.zeros((shape_rows, 3))
    for k in range(m):
        e.reshape(-1)[k] = 1.0
        H[:, k] = op(e).reshape(-1)
        e.reshape(-1)[k] = 0.0
    return H

t0 = time.time()
H = dense(lambda p: sc.hess_vec(y, p, cons), V + 4)
H = 0.5 * (H + H.T)
A0 = dense(lambda p: sc.pad.hess_vec(sc.pad.X, p, cons), V)  # rest-state pad operator (mass, constraints, elastic)
print(f"dense operators in {time.time() - t0:.1f} s; min eig H {np.linalg.eigvalsh(H).min():.3e}")
g = sc.gradient(y, yt, cons, aim).reshape(-1)

node, w, nc = build_coarse_space(sc.pad.X, coarse_grid_dims(sc.pad.X))
P = prolongation_matrix(node, w, nc)  # (3V, 3nc)
print("coarse nodes", nc)
D = sc.diag_blocks(y, cons)[:V]
Dinv = np.linalg.inv(D)
Bfull = H[3 * V:, 3 * V:]
A0c_inv = np.linalg.inv(P.T @ A0 @ P)
Z = np.zeros((n, 3 * nc + 12)); Z[:3 * V, :3 * nc] = P; Z[3 * V:, 3 * nc:] = np.eye(12)
Hc = H.copy(); Hc[:3 * V, :3 * V] -= dense(lambda p: sc.pad.hess_vec(y[:V], p, cons), V)  # contact + ball part of H (pairs, friction, ground, ball)
Ac_c = Z.T @ Hc @ Z; Ac_c[:3 * nc, :3 * nc] += P.T @ A0 @ P
Ac_d = Z.T @ H @ Z

def jac(r):
    z = np.zeros_like(r)
    z[:3 * V] = np.einsum("vij,vj->vi", Dinv, r[:3 * V].reshape(V, 3)).reshape(-1)
    return z
def Ma(r):
    z = jac(r); z[3 * V:] = np.linalg.solve(Bfull, r[3 * V:]); return z
def Mb(r):
    z = Ma(r); z[:3 * V] += P @ (A0c_inv @ (P.T @ r[:3 * V])); return z
def Mz(Ac):
    Ai = np.linalg.inv(Ac)
    return lambda r: jac(r) + Z @ (Ai @ (Z.T @ r))
for name, M in (("a block Jacobi + ball block", Ma), ("b a + rest coarse (kernel)", Mb), ("c coupled coarse, rest pad operator + contact terms", Mz(Ac_c)),
                ("d coupled coarse, exact Galerkin", Mz(Ac_d))):
    for tol in (1e-3, 1e-6):
        d, it = pcg_solve(lambda p: H @ p, M, -g, 4000, tol)
        print(f"{name:55s} tol {tol:.0e}: {it:4d} iterations")
