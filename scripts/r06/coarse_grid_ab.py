"""A/B of the coarse grid of the two-level preconditioner on the C4 pad (8 x 10 x 4 cells): "auto" = (2, 3, 1) cells = 24 nodes against grids
whose nodes coincide with mesh vertices ((4, 5, 1) = 60 nodes, (2, 5, 1) = 36, (4, 5, 0)...), on the ball scene and on C4's scene; 512 envs."""
import sys, time, torch
from tacex_amd.uipc.gelpad_scene import FemBallScene, FemGelpad
from tacex_amd.uipc.uipc_sim import UipcSimCfg
B = 512
for grid in ("auto", (4, 5, 1), (2, 5, 1), (4, 2, 1), (4, 5, 2)):
    for name, make in (("ball", lambda c: FemBallScene(B, "cuda:0", max_newton_iter=64, cfg=c)), ("c4", lambda c: FemGelpad(B, "cuda:0", max_newton_iter=64, cfg=c))):
        cfg = UipcSimCfg(device="cuda:0")
        cfg.linear_system.coarse_grid = grid
        try:
            sc = make(cfg)
        except Exception as ex:
            print(f"{name:5s} grid {grid}: {type(ex).__name__}: {ex}"[:200], flush=True)
            continue
        tot = torch.zeros(4, dtype=torch.float64, device="cuda:0")
        for i in range(21):
            sc.step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mx = 0
        for i in range(21, 84):
            sc.step(i)
            tot += sc.sim.step_info.mean(0)
            mx = max(mx, int(sc.sim.step_info[:, 0].max()))
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 63
        t = tot.cpu().numpy() / 63
        print(f"{name:5s} grid {str(grid):10s}: {ms:7.3f} ms/step  newton {t[0]:.2f} (max {mx})  pcg/newton {t[3] / max(t[0], 1e-9):5.1f}  flags {int(sc.sim.step_info[:, 2].max())}", flush=True)
        del sc
        torch.cuda.empty_cache()
