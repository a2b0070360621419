"""Dump the pre-step state of the envs of the bench's ball scene whose step needs the most PCG iterations per Newton iteration (GPU), for a
replay in the oracle (scripts/r06/ball_replay.py, CPU)."""
import sys, numpy as np, torch
from tacex_amd.uipc.gelpad_scene import FemBallScene
B = 512
sc = FemBallScene(B, "cuda:0", max_newton_iter=64)
sim = sc.sim
keep = []
for i in range(30):
    pre = dict(x=sim.x.clone(), v=sim.v.clone(), q=sim.q.clone(), qv=sim.qv.clone())
    sc.step(i); torch.cuda.synchronize()
    si = sim.step_info
    ratio = si[:, 3] / si[:, 0].clamp_min(1)
    b = int(ratio.argmax())
    print(f"step {i}: worst env {b} newton {int(si[b,0])} pcg {int(si[b,3])}; mean pcg/newton {float(ratio.mean()):.1f}", flush=True)
    if i >= 6:
        keep.append((float(ratio[b]), i, b, {k: t[b].cpu().numpy() for k, t in pre.items()}, sc._aim[b].cpu().numpy(), si[b].cpu().numpy(),
                     sim.x[b].cpu().numpy(), sim.q[b].cpu().numpy()))
keep.sort(key=lambda t: -t[0])
out = {"points": sc.gelpad.points if hasattr(sc.gelpad, "points") else None}
P = np.asarray(sc.gelpad.cfg.mesh_points, np.float64); T = np.asarray(sc.gelpad.cfg.mesh_tets, np.int64)
vb = np.asarray(sc.ball.cfg.mesh_points, np.float64); tb = np.asarray(sc.ball.tris, np.int64)
save = dict(P=P, T=T, vb=vb, tb=tb, back=sc._back.cpu().numpy(), dt=sim.cfg.dt, d_hat=sim.cfg.contact.d_hat, gh=sim.cfg.ground_height,
            init_pos=np.asarray(sc.ball.cfg.init_pos, np.float64))
for k, (r, i, b, pre, aim, si, x1, q1) in enumerate(keep[:4]):
    for name, arr in pre.items():
        save[f"s{k}_{name}"] = arr
    save[f"s{k}_aim"] = aim; save[f"s{k}_info"] = si; save[f"s{k}_x1"] = x1; save[f"s{k}_q1"] = q1; save[f"s{k}_meta"] = np.array([i, b, r])
np.savez_compressed("gpurun_out/r06_ball_states.npz", **save)
print("saved", [(i, b, round(r, 1)) for r, i, b, *_ in keep[:4]])
