import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pathlib import Path
from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
g = np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "fem_meshes.npz")
P = (g["simple_axle_points"] - g["simple_axle_points"].min(0)) * 0.01
T = g["simple_axle_tets"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for streaming in (True, False):
    cfg = UipcSimCfg(device="cuda:0")
    if streaming:
        cfg.linear_system.coarse_grid, cfg.linear_system.vertex_chains = None, None
        cfg.linear_system.deterministic = True
        cfg.contact.enable_friction = False
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
    Pt = torch.from_numpy(P).cuda()
    for i in range(8):
        ind[:, 3] -= depth * sim.contact_gaps().amin(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.step(max_newton_iter=64)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        dent = (Pt[None, :, 2] - sim.x[:, :, 2]).amax(1)
        si = sim.step_info.cpu().numpy()
        print(f"streaming={streaming} step {i}: {el*1e3:.1f} ms, dent min/max {float(dent.min())*1e3:.3f}/{float(dent.max())*1e3:.3f} mm, gap min {float(sim.contact_gaps().amin())*1e3:.3f} mm, "
              f"newton max {si[:,0].max():.0f} mean {si[:,0].mean():.1f}, pcg mean {si[:,3].mean():.0f} max {si[:,3].max():.0f}, flags or {int(np.bitwise_or.reduce(si[:,2].astype(int)))}", flush=True)
