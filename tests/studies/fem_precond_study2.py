"""Second pass of the preconditioner study: iteration counts over SEVERAL states of the C4 trajectory (approach, deep contact,
retraction) for the candidates that are cheap enough to live inside fem_newton_lds_kernel.  python tests/studies/fem_precond_study2.py"""
import math
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from fem_precond_study import assemble, pcg  # noqa: E402
from oracle.fem_oracle import ContactModel, FemModel, contact_distance, newton_step_contact  # noqa: E402
from tacex_amd.uipc.uipc_object import UipcObject, UipcObjectCfg, gelpad_box_mesh  # noqa: E402

nx, ny, nz = 8, 10, 4
P, T = gelpad_box_mesh(nx, ny, nz)
m = FemModel.build(P, T, youngs=1e4, poisson=0.49, density=1e3, dt=0.01, strength=1000.0)
V = len(P)
cons = (P[:, 2] < 1e-9).astype(np.float64)
aim = P.copy(); aim[:, 0] += 0.0002
top = P[:, 2].max(); size = P.max(0); R = 0.004
area = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=m.tets), None).surface_vertex_areas()
z_rest = top + R + 0.0009
cm = ContactModel(area, np.array([1.0, size[0] / 2, size[1] / 2, z_rest, R, 0, 0, 0]), 1e-3, 10e9 * 1e-3, m.dt)


def hats(g, p):
    W = np.zeros((len(p), len(g)))
    for i in range(len(g)):
        w = np.zeros(len(p))
        if i > 0: w = np.where((p >= g[i - 1]) & (p <= g[i]), (p - g[i - 1]) / (g[i] - g[i - 1]), w)
        if i + 1 < len(g): w = np.where((p >= g[i]) & (p <= g[i + 1]), (g[i + 1] - p) / (g[i + 1] - g[i]), w)
        W[:, i] = np.where(np.abs(p - g[i]) < 1e-12, 1.0, w)
    return W


def prolong(cx, cy, cz):
    Wx, Wy, Wz = (hats(np.linspace(0, size[d], n + 1), P[:, d]) for d, n in enumerate((cx, cy, cz)))
    Pn = np.einsum("vi,vj,vk->vijk", Wx, Wy, Wz).reshape(V, -1)
    return sp.kron(sp.csr_matrix(Pn), sp.identity(3)).tocsr()


A0c = assemble(m, P.copy(), cons, None)
coarse = {}
for g in ((2, 2, 1), (2, 3, 1), (4, 5, 1)):
    Pm = prolong(*g)
    coarse[g] = (Pm, np.linalg.inv((Pm.T @ A0c @ Pm).toarray()))
col = np.arange(V) // (nz + 1)
x = P.copy(); v = np.zeros_like(x); depth = 0.0012
print("state | BJ | zline | BJ+c221 BJ+c231 BJ+c451 | zl+c221 zl+c231 zl+c451 | BJ+c451(current op)")
for step in range(16):
    target = z_rest - depth * (0.5 - 0.5 * math.cos(0.3 * (step + 1)))
    d, _ = contact_distance(cm.ind, x); gap = d[area > 0].min()
    cm.ind[3] = max(target, cm.ind[3] - 0.5 * gap) if cm.ind[3] > target else target
    xn = x.copy(); xt = x + m.dt * v + m.dt**2 * np.array([0, 0, -9.8])
    for k in range(8):
        if step in (2, 5, 8, 10, 12, 15) and k in (0, 3):
            A = assemble(m, x, cons, cm)
            b = -(m.gradient(x, xt, cons, aim) + cm.gradient(x)).reshape(-1)
            Dm = sp.block_diag([np.linalg.inv(A[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) for i in range(V)]).tocsr()
            blocks = []
            for c in range(col.max() + 1):
                dd = (np.where(col == c)[0][:, None] * 3 + np.arange(3)).reshape(-1)
                blocks.append((dd, np.linalg.inv(A[dd][:, dd].toarray())))
            def zline(r):
                z = np.zeros_like(r)
                for dd, Bi in blocks: z[dd] = Bi @ r[dd]
                return z
            res = [pcg(A, b, lambda r: Dm @ r)[1], pcg(A, b, zline)[1]]
            for g in coarse:
                Pm, Aci = coarse[g]
                res.append(pcg(A, b, lambda r: Dm @ r + Pm @ (Aci @ (Pm.T @ r)))[1])
            for g in coarse:
                Pm, Aci = coarse[g]
                res.append(pcg(A, b, lambda r: zline(r) + Pm @ (Aci @ (Pm.T @ r)))[1])
            Pm = coarse[(4, 5, 1)][0]
            Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
            res.append(pcg(A, b, lambda r: Dm @ r + Pm @ (Aci @ (Pm.T @ r)))[1])
            dmin = contact_distance(cm.ind, x)[0][area > 0].min()
            print(f"step {step} newton {k} gap {dmin * 1e3:.3f} mm |", *res)
        x, st = newton_step_contact(m, cm, x, xt, cons, aim, pcg_max_iter=2000, ls_max_iter=40)
    v = (x - xn) / m.dt
