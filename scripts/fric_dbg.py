import sys
from pathlib import Path
import numpy as np, torch
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from test_fem_gpu import _c4_scene
from oracle.fem_oracle import fem_step, FrictionModel, newton_step_contact
B = 1
sim, m, P, cons, aim, cms = _c4_scene(B)
sim.cfg.newton.velocity_tol = 2e-3
sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-6
xo, vo = P.copy(), np.zeros_like(P)
ind = sim.contact_indenters
prev = ind[:, 1:4].cpu().numpy().copy()
for k in range(3):
    ind[:, 1] += 2e-5
    gap = sim.contact_gaps().amin(1)
    ind[:, 3] -= 0.4 * gap
    cur = ind[:, 1:4].cpu().numpy().copy()
    disp = cur - prev if k > 0 else np.zeros_like(cur)
    prev = cur
    cms[0].ind[1:4] = cur[0]
    # one Newton iteration at a time on the GPU is not possible with friction; compare the whole step with 1, 2, 3 ... iterations
    x_start, v_start = sim.x.clone(), sim.v.clone()
    for nmax in (1, 2, 3, 12):
        sim.x.copy_(x_start); sim.v.copy_(v_start)
        # undo the library's indenter bookkeeping: it advanced ind_prev at the end of the previous call
        sim.step(max_newton_iter=nmax)
        si = sim.step_info.cpu().numpy()[0]; st = sim.stats.cpu().numpy()[0]
        x1, v1, io = fem_step(m, cms[0], xo, vo, cons, aim[0], gravity=sim.cfg.gravity, max_newton=nmax, velocity_tol=2e-3, pcg_max_iter=600,
                              pcg_tol_rate=1e-6, coarse=sim.coarse_space, friction=(0.5, 0.01, disp[0]))
        print(f"step {k} nmax {nmax}: gpu newton {si[0]:.0f} pcg {si[3]:.0f} stats {st} | oracle {io} | max |dx| {np.abs(sim.x[0].cpu().numpy() - x1).max():.3e}")
        if nmax < 12:
            # restore ind_prev semantics: re-running the same step must see the same displacement -> reset through set_contact? not available; break after first
            pass
    xo, vo = x1, v1
