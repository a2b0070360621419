"""simple_axle scene of bench.py::fem_axle_entry stepped with the PCG tolerance at its default (1e-3 on r.z) and a million times tighter: do the states agree
within the Newton tolerance?"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pathlib import Path
from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
g = np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "fem_meshes.npz")
P = (g["simple_axle_points"] - g["simple_axle_points"].min(0)) * 0.01
T = g["simple_axle_tets"]
B = 64
out = {}
for tol in (1e-3, 1e-9):
    cfg = UipcSimCfg(device="cuda:0")
    cfg.newton.velocity_tol = 2e-3
    cfg.linear_system.tol_rate = tol
    cfg.linear_system.max_iter = 4000
    sim = UipcSim(cfg, num_envs=B)
    UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
    sim.setup_sim(constraint_strength_ratio=1000.0)
    ends = np.where((P[:, 0] < 0.002) | (P[:, 0] > P[:, 0].max() - 0.002))[0]
    sim.set_constraints(ends, torch.from_numpy(np.repeat(P[None, ends], B, 0)).cuda())
    ind = torch.zeros((B, 8), dtype=torch.float64, device="cuda:0")
    ind[:, 0], ind[:, 1], ind[:, 2], ind[:, 4] = 1.0, P[:, 0].max() / 2, P[:, 1].max() / 2, 0.004
    ind[:, 3] = P[:, 2].max() + 0.004 + 0.0009
    sim.set_contact_indenters(ind)
    ind = sim.contact_indenters
    depth = torch.linspace(0.2, 0.4, B, device="cuda:0", dtype=torch.float64)
    # the SAME indenter trajectory in both runs (prescribed, not gap-driven): press 60 um per step, slide back and forth
    tot = [0, 0]
    for i in range(12):
        ind[:, 3] -= 6e-5 * depth / 0.3
        ind[:, 1] += 2e-5 * (1 if (i // 4) % 2 == 0 else -1)
        sim.step(max_newton_iter=200)
        info = sim.check_step()
        tot[0] += int(info["newton_iters"].max()); tot[1] += int(info["pcg_iters"].max())
        assert len(info["line_search_failed_envs"]) == 0, (tol, i)
    out[tol] = sim.x.cpu().numpy()
    Pt = torch.from_numpy(P).cuda()
    print(f"tol_rate {tol:g}: dent {float((Pt[None, :, 2] - sim.x[:, :, 2]).amax()) * 1e3:.3f} mm, smallest gap {float(sim.contact_gaps().amin()) * 1e3:.3f} mm, worst env per step summed: Newton {tot[0]}, PCG {tot[1]}", flush=True)
d = np.abs(out[1e-3] - out[1e-9]).max(axis=(1, 2))
print(f"max |x(1e-3) - x(1e-9)| over envs: median {np.median(d) * 1e6:.2f} um, max {d.max() * 1e6:.2f} um  (Newton tolerance per step: 20 um)")
