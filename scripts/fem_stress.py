"""Long-run check of the gelpad FEM step: the C4 scene for many periods of the indenter's motion (both motions), statistics of what the
solver reported - no penetration, no failed line search, no env at the iteration cap, finite state - and per-step latency quantiles.
usage: python scripts/fem_stress.py [steps=420] [envs=512]"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.uipc.gelpad_scene import FemGelpad
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 420
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for motion in ("breathing", "rolling"):
    fem = FemGelpad(B, "cuda:0", max_newton_iter=64, motion=motion)
    fem.ms_log = []
    fem.info_sum = torch.zeros(4, dtype=torch.float64, device="cuda:0")
    flags_or = torch.zeros(B, dtype=torch.int64, device="cuda:0")
    it_max = torch.zeros((), dtype=torch.float64, device="cuda:0")
    gap_min = torch.full((), float("inf"), dtype=torch.float64, device="cuda:0")
    ls_events = torch.zeros(steps, dtype=torch.float64, device="cuda:0")
    pcg_max = torch.zeros(steps, dtype=torch.float64, device="cuda:0")
    pcg_arg = torch.zeros(steps, dtype=torch.int64, device="cuda:0")
    for i in range(steps):
        fem.step(i)
        si = fem.sim.step_info
        flags_or |= si[:, 2].to(torch.int64)
        it_max = torch.maximum(it_max, si[:, 0].max())
        gap_min = torch.minimum(gap_min, fem.sim.contact_gaps().amin())
        ls_events[i] = (si[:, 2].to(torch.int64) & 2).ne(0).sum()
        pcg_max[i] = si[:, 3].max()
        pcg_arg[i] = si[:, 3].argmax()
    fem.flush()
    torch.cuda.synchronize()
    ms = np.array(fem.ms_log[21:])
    x = fem.sim.x
    fl = flags_or.cpu().numpy().astype(int)
    worst = int(pcg_max.argmax())
    slow = 21 + int(np.argmax(fem.ms_log[21:]))
    print(f"{motion}: most PCG iterations of an env in one step: {int(pcg_max[worst])} (step {worst}, env {int(pcg_arg[worst])}); slowest step {slow}: {fem.ms_log[slow]:.2f} ms, "
          f"its worst env {int(pcg_arg[slow])} with {int(pcg_max[slow])} PCG iterations", flush=True)
    print(f"{motion}: {steps} steps x {B} envs: finite {bool(torch.isfinite(x).all())}, smallest gap of any step {float(gap_min) * 1e3:.4f} mm, "
          f"envs ever flagged penetration {int((fl & 1).astype(bool).sum())}, line search {int((fl & 2).astype(bool).sum())}, coarse correction dropped {int((fl & 4).astype(bool).sum())}, PSD-safe mode {int((fl & 8).astype(bool).sum())} ({int(ls_events.sum())} env-steps of {steps * B}, steps {[int(v) for v in torch.nonzero(ls_events).flatten()[:12].cpu()]}), max Newton iterations {int(it_max)}; "
          f"ms per step mean {ms.mean():.3f} median {np.median(ms):.3f} p90 {np.quantile(ms, 0.9):.3f} max {ms.max():.3f}", flush=True)
