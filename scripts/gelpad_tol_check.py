"""C4 gelpad scene stepped with the PCG tolerance at its default (1e-3 on r.z) and much tighter: how far apart are the states after a period?"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tacex_amd.uipc.gelpad_scene import FemGelpad
out = {}
for tol in (1e-3, 1e-6, 1e-10):
    fem = FemGelpad(512, "cuda:0", max_newton_iter=64)
    fem.sim.cfg.linear_system.tol_rate = tol
    newton = pcg = 0
    for i in range(42):
        fem.step(i)
        si = fem.sim.step_info
        newton += float(si[:, 0].mean()); pcg += float(si[:, 3].mean())
    torch.cuda.synchronize()
    out[tol] = fem.sim.x.cpu().numpy()
    P = torch.from_numpy(fem.gelpad.points).cuda()
    print(f"tol_rate {tol:g}: Newton iterations per env and step {newton / 42:.2f}, PCG {pcg / 42:.1f}; dent max {float((P[None, :, 2] - fem.sim.x[:, :, 2]).amax()) * 1e3:.3f} mm", flush=True)
for tol in (1e-3, 1e-6):
    d = np.abs(out[tol] - out[1e-10]).max(axis=(1, 2))
    print(f"max |x({tol:g}) - x(1e-10)| over envs after 42 steps: median {np.median(d) * 1e6:.2f} um, max {d.max() * 1e6:.2f} um  (Newton tolerance per step: 500 um)")
