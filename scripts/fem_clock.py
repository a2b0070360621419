#!/usr/bin/env python3
"""Section clock of one PCG iteration of fem_newton_lds_kernel (debug builds -DTACEX_FEM_CLOCK=g, g = 0..2: four counters each).
usage: TACEX_LIB_TAG=fc<g> python scripts/fem_clock.py <g> [B]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tacex_amd.uipc import UipcSim, UipcSimCfg, UipcObject, UipcObjectCfg
from tacex_amd.uipc.uipc_object import gelpad_box_mesh
g = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
P, T = gelpad_box_mesh(8, 10, 4)
sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
obj = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
sim.setup_sim()
top = np.where(P[:, 2] > P[:, 2].max() - 1e-9)[0]
aim = torch.from_numpy(P[top]).cuda()[None].repeat(B, 1, 1)
aim[:, :, 2] -= torch.linspace(0.0002, 0.0012, B, device="cuda", dtype=torch.float64)[:, None]
sim.set_constraints(top, aim)
sim.x_tilde = sim.x.clone()
x0 = sim.x.clone()
for _ in range(3):
    sim.x.copy_(x0); sim.newton_step()
torch.cuda.synchronize()
st = sim.stats.cpu().numpy()
names = ["sweep", "Hp+pHp sum", "update+BJ (first call: garbage)", "rs+2 syncs", "restrict+sync", "rc sum+sync", "coarse solve+sync", "prolong", "sweep: make+hv write", "sweep: barrier", "sweep: csr gather", "sweep: 2nd barrier"]
if g == 3:
    print("group 3 (cycles per Newton iteration):", dict(zip(["gradient + contact", "block assembly + chain factor", "PCG (%d iterations)" % 10, "line search + update"], [float(st[:, k].mean()) for k in range(4)])))
else:
    print("group", g, {names[4 * g + k]: float(st[:, k].mean()) for k in range(4)})
