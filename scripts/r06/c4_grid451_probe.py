"""C4 scene with the (4, 5, 1) coarse grid: which env / step runs into the Newton cap, and with what flags (scripts/r06/coarse_grid_ab.py saw max 64)."""
import torch
from tacex_amd.uipc.gelpad_scene import FemGelpad
from tacex_amd.uipc.uipc_sim import UipcSimCfg
cfg = UipcSimCfg(device="cuda:0")
cfg.linear_system.coarse_grid = (4, 5, 1)
sc = FemGelpad(512, "cuda:0", max_newton_iter=64, cfg=cfg)
for i in range(84):
    sc.step(i); torch.cuda.synchronize()
    si = sc.sim.step_info
    b = int(si[:, 0].argmax())
    if int(si[b, 0]) >= 8:
        print(f"step {i}: env {b} newton {int(si[b,0])} max|d| {float(si[b,1]):.3e} flags {int(si[b,2])} pcg {int(si[b,3])}; envs >= 8 iterations: {int((si[:,0] >= 8).sum())}", flush=True)
print("done; flags over all envs at the last step:", int(sc.sim.step_info[:, 2].max()))
