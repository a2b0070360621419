"""c4_ball scene probe: per-step Newton / PCG counts, flags and time of FemBallScene (default tolerances), 64 envs."""
import sys, time, torch
from tacex_amd.uipc.gelpad_scene import FemBallScene
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
import os
from tacex_amd.uipc.uipc_sim import UipcSimCfg
cfg = None
if os.environ.get("TACEX_BALL_EDGE_EDGE") == "0":  # A/B: point-triangle pairs alone
    cfg = UipcSimCfg(device="cuda:0")
    cfg.contact.edge_edge = False
if "TACEX_BALL_LS_REFINE" in os.environ:  # A/B: bisections after a cut line search (cfg.line_search.refine, default 4)
    cfg = cfg or UipcSimCfg(device="cuda:0")
    cfg.linear_system.coarse_grid = (4, 5, 1)  # (what FemBallScene sets when it builds its own cfg)
    cfg.line_search.refine = int(os.environ["TACEX_BALL_LS_REFINE"])
sc = FemBallScene(B, "cuda:0", max_newton_iter=64, cfg=cfg)
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.step(i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    si = sc.sim.step_info
    print(f"step {i:2d} {ms:8.2f} ms newton mean {float(si[:,0].mean()):5.1f} max {int(si[:,0].max()):3d} pcg/newton {float((si[:,3]/si[:,0].clamp_min(1)).mean()):6.1f} "
          f"flags {int(si[:,2].max())} ball z-z0 [{float((sc.sim.q[:,0,2]).min()-0.0105)*1e6:8.1f}, {float((sc.sim.q[:,0,2]).max()-0.0105)*1e6:8.1f}] um", flush=True)
