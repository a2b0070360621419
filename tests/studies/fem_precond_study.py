"""Preconditioner study for the gelpad Newton system (CPU, NumPy/SciPy on the FEM oracle): PCG iteration counts of candidate
preconditioners on the C4 scene (8 x 10 x 4 gelpad, back face attached, sphere indenter in contact) - the numbers that chose the
two-level preconditioner of fem_newton_lds_kernel.  Run: python tests/studies/fem_precond_study.py"""
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle.fem_oracle import ContactModel, FemModel, newton_step_contact  # noqa: E402
from tacex_amd.uipc.uipc_object import gelpad_box_mesh  # noqa: E402


def assemble(m, x, cons, cm):
    He = m.element_hessian(x) * m.dt**2
    V = len(m.X)
    rows, cols, vals = [], [], []
    dof = (m.tets[:, :, None] * 3 + np.arange(3)).reshape(len(m.tets), 12)
    rows = np.repeat(dof, 12, axis=1).reshape(-1)
    cols = np.tile(dof, (1, 12)).reshape(-1)
    A = sp.coo_matrix((He.reshape(-1), (rows, cols)), shape=(3 * V, 3 * V)).tocsr()
    md = m.mass * (1.0 + m.strength * cons)
    A = A + sp.diags(np.repeat(md, 3))
    if cm is not None:
        Hb = cm.hess_blocks(x)
        A = A + sp.block_diag([Hb[v] for v in range(V)]).tocsr()
    return A.tocsr()


def pcg(A, b, prec, tol=1e-3, maxit=2000):
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; rz0 = rz; it = 0
    while it < maxit and rz > tol * tol * rz0:
        Ap = A @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap; z = prec(r); rzn = r @ z; p = z + rzn / rz * p; rz = rzn; it += 1
    return x, it


NSTEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 14


def main():
    nx, ny, nz = 8, 10, 4
    P, T = gelpad_box_mesh(nx, ny, nz)
    m = FemModel.build(P, T, youngs=1e4, poisson=0.49, density=1e3, dt=0.01, strength=1000.0)
    V = len(P)
    cons = (P[:, 2] < 1e-9).astype(np.float64)
    aim = P.copy(); aim[:, 0] += 0.0002
    top = P[:, 2].max(); size = P.max(0)
    R = 0.004
    from tacex_amd.uipc.uipc_object import UipcObject, UipcObjectCfg
    area = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=m.tets), None).surface_vertex_areas()
    z_rest = top + R + 0.0009
    ind = np.array([1.0, size[0] / 2, size[1] / 2, z_rest, R, 0, 0, 0])
    cm = ContactModel(area, ind, 1e-3, 10e9 * 1e-3, m.dt)
    from oracle.fem_oracle import contact_distance
    x = P.copy(); v = np.zeros_like(x)
    depth = 0.0012
    for step in range(NSTEPS):  # the bench's breathing trajectory: the indenter moves at most half the current gap per step
        import math
        target = z_rest - depth * (0.5 - 0.5 * math.cos(0.3 * (step + 1)))
        d, _ = contact_distance(cm.ind, x)
        gap = d[area > 0].min()
        cm.ind[3] = max(target, cm.ind[3] - 0.5 * gap) if cm.ind[3] > target else target
        xn = x.copy(); xt = x + m.dt * v + m.dt**2 * np.array([0, 0, -9.8])
        its = []
        for k in range(8):
            x, st = newton_step_contact(m, cm, x, xt, cons, aim, pcg_max_iter=2000)
            its.append(int(st[3]))
        v = (x - xn) / m.dt
        print("step", step, "z", cm.ind[3] - z_rest, "pcg iters", its, "step", st[2])
    xt = x + m.dt * v + m.dt**2 * np.array([0, 0, -9.8])
    A = assemble(m, x, cons, cm)
    g = m.gradient(x, xt, cons, aim) + cm.gradient(x)
    b = -g.reshape(-1)
    # block Jacobi
    D = [np.linalg.inv(A[3 * v:3 * v + 3, 3 * v:3 * v + 3].toarray()) for v in range(V)]
    Dm = sp.block_diag(D).tocsr()
    _, it = pcg(A, b, lambda r: Dm @ r); print("block-Jacobi:", it)
    # z-line block Jacobi: one block per (i, j) column of the grid (nz + 1 vertices = 15 dofs)
    col = (np.arange(V) // (nz + 1))
    blocks = []
    perm = []
    for c in range(col.max() + 1):
        vs = np.where(col == c)[0]
        dd = (vs[:, None] * 3 + np.arange(3)).reshape(-1)
        blocks.append(np.linalg.inv(A[dd][:, dd].toarray())); perm.append(dd)
    def zline(r):
        z = np.zeros_like(r)
        for dd, Bi in zip(perm, blocks): z[dd] = Bi @ r[dd]
        return z
    _, it = pcg(A, b, zline); print("z-line block Jacobi (15x15):", it)
    # x-line (i-direction rows of nx+1 vertices) for comparison
    # two-level: trilinear coarse grid
    def coarse(cx, cy, cz, Aop):
        gx, gy, gz = np.linspace(0, size[0], cx + 1), np.linspace(0, size[1], cy + 1), np.linspace(0, size[2], cz + 1)
        def hat(g, p):
            W = np.zeros((len(p), len(g)))
            for i in range(len(g)):
                l = g[i - 1] if i > 0 else None; c = g[i]; r = g[i + 1] if i + 1 < len(g) else None
                w = np.zeros(len(p))
                if l is not None: w = np.where((p >= l) & (p <= c), (p - l) / (c - l), w)
                if r is not None: w = np.where((p >= c) & (p <= r), (r - p) / (r - c), w)
                w = np.where(np.abs(p - c) < 1e-12, 1.0, w)
                W[:, i] = w
            return W
        Wx, Wy, Wz = hat(gx, P[:, 0]), hat(gy, P[:, 1]), hat(gz, P[:, 2])
        Pn = np.einsum("vi,vj,vk->vijk", Wx, Wy, Wz).reshape(V, -1)
        Pm = sp.kron(sp.csr_matrix(Pn), sp.identity(3)).tocsr()
        Ac = (Pm.T @ Aop @ Pm).toarray()
        return Pm, np.linalg.inv(Ac)
    A0 = assemble(m, P.copy(), cons * 0, None)   # rest state, no constraints, no contact: a CONSTANT matrix
    A0c = assemble(m, P.copy(), cons, None)      # rest state with the constraint masses
    for (cx, cy, cz) in ((2, 2, 1), (4, 5, 1), (2, 3, 1), (8, 10, 1)):
        for name, Aop in (("current", A), ("rest", A0), ("rest+cons", A0c)):
            Pm, Aci = coarse(cx, cy, cz, Aop)
            prec = lambda r: Dm @ r + Pm @ (Aci @ (Pm.T @ r))
            _, it = pcg(A, b, prec)
            # multiplicative (symmetric): jacobi pre, coarse, jacobi post
            def mult(r):
                w = 0.6
                z = w * (Dm @ r)
                z = z + Pm @ (Aci @ (Pm.T @ (r - A @ z)))
                z = z + w * (Dm @ (r - A @ z))
                return z
            _, it2 = pcg(A, b, mult)
            _, it3 = pcg(A, b, lambda r: zline(r) + Pm @ (Aci @ (Pm.T @ r)))
            print(f"coarse {cx}x{cy}x{cz} ({Pm.shape[1]} dofs) [{name}]: additive {it}, multiplicative(3 matvec) {it2}, z-line + coarse additive {it3}")


if __name__ == "__main__":
    main()
