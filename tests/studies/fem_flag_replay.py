"""CPU replay of the envs scripts/fem_flag_dump.py caught with a solver flag: the same time step through the oracle's fem_step.
usage: python tests/studies/fem_flag_replay.py gpurun_out/r06b/flags.npz"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle.fem_oracle import ContactModel, FemModel, chain_tables, fem_step
from tacex_amd.uipc.coarse_space import build_vertex_chains
from tacex_amd.uipc.uipc_object import UipcObject, UipcObjectCfg, gelpad_box_mesh

d = np.load(sys.argv[1])
P, T = gelpad_box_mesh(8, 10, 4)
obj = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T))
m = FemModel.build(P, T, youngs=obj.cfg.constitution_cfg.youngs_modulus * 1e6, poisson=obj.cfg.constitution_cfg.poisson_rate,
                   density=obj.cfg.mass_density, dt=0.01, strength=1000.0)
area = obj.surface_vertex_areas()
coarse = (d["coarse_node"], d["coarse_w"], d["coarse_aci"])
chains = chain_tables([list(map(int, c)) for c in build_vertex_chains(P, T) if len(c) > 1], len(P))
for ev in range(int(d["events"])):
    for k, env in enumerate(d[f"e{ev}_envs"]):
        cm = ContactModel(area, d[f"e{ev}_ind"][k].copy(), 1e-3, 10.0 * 1e9 * 1e-3, m.dt)
        disp = d[f"e{ev}_ind"][k][1:4] - d[f"e{ev}_ind_prev"][k][1:4]
        # (ind_prev = the indenter of the step before the previous one as dumped; the displacement the kernel saw is ind - ind_before)
        disp = d[f"e{ev}_ind"][k][1:4] - d[f"e{ev}_ind_before"][k][1:4]
        x, v, io = fem_step(m, cm, d[f"e{ev}_x"][k], d[f"e{ev}_v"][k], d[f"e{ev}_cons"][k].astype(np.float64), d[f"e{ev}_aim"][k],
                            max_newton=64, velocity_tol=0.05, pcg_max_iter=1024, pcg_tol_rate=1e-3, coarse=coarse, chains=chains,
                            friction=(0.5, 0.01, disp))
        print(f"step {int(d[f'e{ev}_step'])} env {env}: GPU step_info {d[f'e{ev}_info'][k].tolist()} | oracle {io.tolist()} | "
              f"max |x - x_gpu| {np.abs(x - d[f'e{ev}_x_after'][k]).max():.2e}", flush=True)
