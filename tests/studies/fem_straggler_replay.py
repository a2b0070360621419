"""CPU replay of one env of the C4 scene through the oracle, one Newton iteration at a time (why do some envs need 20-30 iterations?).
usage: python tests/studies/fem_straggler_replay.py gpurun_out/r04i/env_s11.npz <k = index into the dumped envs> [max_iters]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle.fem_oracle import ContactModel, FemModel, FrictionModel, chain_tables, contact_distance, newton_step_contact
from tacex_amd.uipc.coarse_space import build_vertex_chains
from tacex_amd.uipc.uipc_object import UipcObject, UipcObjectCfg, gelpad_box_mesh

d = np.load(sys.argv[1])
k = int(sys.argv[2]); nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 40
P, T = gelpad_box_mesh(8, 10, 4)
obj = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T))
m = FemModel.build(P, T, youngs=obj.cfg.constitution_cfg.youngs_modulus * 1e6, poisson=obj.cfg.constitution_cfg.poisson_rate,
                   density=obj.cfg.mass_density, dt=0.01, strength=1000.0)
area = obj.surface_vertex_areas()
dhat, kappa = 1e-3, 10.0 * 1e9 * 1e-3
ind = d["ind"][k].copy()
cm = ContactModel(area, ind, dhat, kappa, m.dt)
cons = d["cons"][k].astype(np.float64); aim = d["aim"][k]
x, v = d["x"][k].copy(), d["v"][k].copy()
coarse = (d["coarse_node"], d["coarse_w"], d["coarse_aci"])
chains = chain_tables([list(map(int, c)) for c in build_vertex_chains(P, T) if len(c) > 1], len(P))
print("env", d["envs"][k], "step", d["step"], "GPU step_info", d["step_info"][k], "indenter moved by", d["ind"][k][1:4] - d["ind_before"][k][1:4])
xt = x + m.dt * v + m.dt**2 * np.array([0.0, 0.0, -9.8])
tol = 0.05 * m.dt
d0 = None
for it in range(nmax):
    g0, n0 = contact_distance(cm.ind, x, cm.mesh)
    surf = area > 0
    act = surf & (g0 < dhat)
    x_new, st, dd = newton_step_contact(m, cm, x, xt, cons, aim, 1024, 1e-3, 8, coarse, d0, True, None, chains)
    d0 = (1.0 - st[2]) * dd if 0.0 < st[2] < 1.0 else None
    g1, _ = contact_distance(cm.ind, x_new, cm.mesh)
    act1 = surf & (g1 < dhat)
    print(f"it {it:2d}: E0 {st[0]:.6e} dE {st[1]-st[0]:+.2e} step {st[2]:.3e} (ccd {st[5]:.2e}) pcg {int(st[3]):3d} max|d| {st[4]:.2e} | active {act.sum():3d} -> {act1.sum():3d} "
          f"(new {int((act1 & ~act).sum())}, left {int((act & ~act1).sum())}) | min gap {g0[surf].min()/dhat:.4f} -> {g1[surf].min()/dhat:.4f} | "
          f"|d| of new-zone vertices max {np.abs(dd[act1 & ~act]).max() if (act1 & ~act).any() else 0:.2e}")
    x = x_new
    if st[4] <= tol:
        print("converged"); break

# ---- the same step through fem_step with the friction phase on (what the GPU ran) ----
import oracle.fem_oracle as fo
disp = d["ind"][k][1:4] - d["ind_before"][k][1:4]
orig = fo.newton_step_contact
trace = []
def traced(*a, **kw):
    r = orig(*a, **kw)
    st = r[1]
    fr = a[12] if len(a) > 12 else kw.get("fr")
    trace.append((st[2], int(st[3]), st[4], st[5], fr is not None))
    return r
fo.newton_step_contact = traced
x2, v2, info = fo.fem_step(m, cm, d["x"][k].copy(), d["v"][k].copy(), cons, aim, max_newton=nmax, velocity_tol=0.05, pcg_max_iter=1024, pcg_tol_rate=1e-3,
                           coarse=coarse, chains=chains, friction=(0.5, 0.01, disp))
print("fem_step with friction:", info)
for i, t in enumerate(trace):
    print(f"  it {i:2d}: step {t[0]:.3e} (ccd {t[3]:.2e}) pcg {t[1]:3d} max|d| {t[2]:.2e} friction phase {t[4]}")
print("GPU vs oracle end state max |dx|:", np.abs(d["x_after"][k] - x2).max())
