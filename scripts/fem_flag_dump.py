"""Run the C4 gelpad scene and dump the pre-step state of envs whose step ended with a flag (line search / penetration), for a CPU replay.
usage: python scripts/fem_flag_dump.py <steps> <out.npz> [motion=breathing] [max_events=3]"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.uipc.gelpad_scene import FemGelpad
steps, out = int(sys.argv[1]), sys.argv[2]
motion = sys.argv[3] if len(sys.argv) > 3 else "breathing"
max_events = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fem = FemGelpad(512, "cuda:0", max_newton_iter=64, motion=motion)
d, ev = {}, 0
ind_prev = fem.ind.clone()
for i in range(steps):
    x0, v0, ind0 = fem.sim.x.clone(), fem.sim.v.clone(), fem.ind.clone()
    fem.step(i)
    si = fem.sim.step_info.cpu().numpy()
    fl = np.where(si[:, 2] != 0)[0]
    if len(fl) and ev < max_events:
        e = fl[:4].tolist()
        print(f"step {i}: {len(fl)} flagged envs {fl.tolist()[:12]}; dumping {e}: step_info {si[e].tolist()}", flush=True)
        d.update({f"e{ev}_x": x0[e].cpu().numpy(), f"e{ev}_v": v0[e].cpu().numpy(), f"e{ev}_ind_prev": ind_prev[e].cpu().numpy(),
                  f"e{ev}_ind_before": ind0[e].cpu().numpy(), f"e{ev}_ind": fem.ind[e].cpu().numpy(), f"e{ev}_aim": fem.sim.aim_position[e].cpu().numpy(),
                  f"e{ev}_cons": fem.sim.is_constrained[e].cpu().numpy(), f"e{ev}_x_after": fem.sim.x[e].cpu().numpy(), f"e{ev}_info": si[e],
                  f"e{ev}_envs": np.array(e), f"e{ev}_step": i})
        ev += 1
    ind_prev = ind0  # the indenter the PREVIOUS step was solved with = what this step's displacement is measured from
d.update(coarse_node=fem.sim.coarse_space[0], coarse_w=fem.sim.coarse_space[1], coarse_aci=fem.sim.coarse_space[2], events=ev)
np.savez(out, **d)
print("events", ev)
