"""Dump the state of chosen envs of the C4 gelpad scene right before a chosen step (for a CPU replay through the oracle).
usage: python scripts/fem_dump_env.py <step> <env,env,...> <out.npz>   (env vars NEWTON_CAP, FEM_MOTION)"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.uipc.gelpad_scene import FemGelpad
step, envs, out = int(sys.argv[1]), [int(e) for e in sys.argv[2].split(",")], sys.argv[3]
fem = FemGelpad(512, "cuda:0", max_newton_iter=int(os.environ.get("NEWTON_CAP", "64")), motion=os.environ.get("FEM_MOTION", "breathing"))
d = {}
for i in range(step + 1):
    if i == step:
        torch.cuda.synchronize()
        d.update(x=fem.sim.x[envs].cpu().numpy(), v=fem.sim.v[envs].cpu().numpy(), ind_before=fem.ind[envs].cpu().numpy())
    fem.step(i)
    if i == step - 1:
        d["ind_prev"] = fem.ind[envs].cpu().numpy()
torch.cuda.synchronize()
si = fem.sim.step_info.cpu().numpy()
d.update(ind=fem.ind[envs].cpu().numpy(), aim=fem.sim.aim_position[envs].cpu().numpy(), cons=fem.sim.is_constrained[envs].cpu().numpy(),
         x_after=fem.sim.x[envs].cpu().numpy(), step_info=si[envs], envs=np.array(envs), step=step,
         coarse_node=fem.sim.coarse_space[0], coarse_w=fem.sim.coarse_space[1], coarse_aci=fem.sim.coarse_space[2])
np.savez(out, **d)
print("step_info of the dumped envs:", si[envs])
