import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from test_fem_gpu import _c4_scene, _chains
from oracle.fem_oracle import fem_step
PRESS=float(sys.argv[1]) if len(sys.argv)>1 else 0.45
MAXN=int(sys.argv[2]) if len(sys.argv)>2 else 12
NPRESS=int(sys.argv[3]) if len(sys.argv)>3 else 4
for mu in (0.5,):
    sim, m, P, cons, aim, cms = _c4_scene(1)
    sim.cfg.contact.default_friction_ratio = mu
    sim.cfg.contact.enable_friction = mu > 0
    sim.cfg.newton.velocity_tol = 2e-3
    sim.cfg.linear_system.max_iter, sim.cfg.linear_system.tol_rate = 600, 1e-6
    sim.set_contact_indenters(sim.contact_indenters)
    ind = sim.contact_indenters
    xo, vo = P.copy(), np.zeros_like(P)
    prev = None
    for k in range(NPRESS+4):
        if k >= NPRESS: ind[:, 1] += 1e-4
        gap = float(sim.contact_gaps().amin())
        if k < NPRESS: ind[:, 3] -= PRESS * gap
        elif gap < 2e-4: ind[:, 3] += 2e-4 - gap
        cur = ind[0, 1:4].cpu().numpy().copy()
        disp = cur - prev if prev is not None else np.zeros(3)
        prev = cur
        cms[0].ind[1:4] = cur
        sim.step(max_newton_iter=MAXN)
        info = sim.check_step()
        xo, vo, io = fem_step(m, cms[0], xo, vo, cons, aim[0], gravity=sim.cfg.gravity, max_newton=MAXN, velocity_tol=2e-3, pcg_max_iter=600,
                              pcg_tol_rate=1e-6, coarse=sim.coarse_space, chains=_chains(sim), friction=(mu, sim.cfg.contact.eps_velocity, disp))
        x = sim.x[0].cpu().numpy()
        print(k, "gap %.3f mm"%(gap*1e3), "gpu newton", sim.last_newton_iters.cpu().numpy() if hasattr(sim.last_newton_iters,'cpu') else sim.last_newton_iters, "oracle", io, "max|x-xo| %.2e"%np.abs(x-xo).max(), flush=True)
