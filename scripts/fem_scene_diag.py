"""Per-step solver statistics of the C4 gelpad scene (which envs / steps cost what)."""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.uipc.gelpad_scene import FemGelpad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fem = FemGelpad(B, "cuda:0", max_newton_iter=int(os.environ.get("NEWTON_CAP", "8")), motion=os.environ.get("FEM_MOTION", "breathing"))

if os.environ.get("NOFRIC"):
    fem.sim.cfg.contact.enable_friction = False
    fem.sim.set_contact_indenters(fem.sim.contact_indenters)
    fem.ind = fem.sim.contact_indenters
fem.ms_log = []
for i in range(32):
    fem.step(i)
    torch.cuda.synchronize()
    si = fem.sim.step_info.cpu().numpy()
    st = fem.sim.stats.cpu().numpy()
    ms = fem.fem_ms_last()
    worst = int(np.argmax(si[:, 3]))
    print(f"step {i:2d}: {ms:7.3f} ms | newton mean {si[:,0].mean():.2f} max {si[:,0].max():.0f} (#at cap {int((si[:,0]>=fem.max_newton_iter).sum())}) | pcg total mean {si[:,3].mean():.0f} "
          f"max {si[:,3].max():.0f} (env {worst}) | last-iter step min {st[:,2].min():.2e} | flags ls {int((si[:,2].astype(int)&2).astype(bool).sum())} "
          f"| gap min {float(fem.sim.contact_gaps().amin())*1e3:.3f} mm | penetration flags {int((si[:,2].astype(int)&1).sum())}", flush=True)
