import sys, math
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.uipc.gelpad_scene import FemGelpad
fem = FemGelpad(512, "cuda:0")
for i in range(31):
    if i >= 27:
        gap0 = fem.sim.contact_gaps().amin(1).clone(); z0 = fem.ind[:, 3].clone()
    fem.step(i)
    torch.cuda.synchronize()
    if i >= 27:
        si = fem.sim.step_info.cpu().numpy(); st = fem.sim.stats.cpu().numpy()
        fl = np.nonzero(si[:, 2].astype(int) & 1)[0]
        gap1 = fem.sim.contact_gaps().amin(1)
        print("step", i, "flagged", fl[:10], "n", len(fl))
        for b in fl[:4]:
            print("  env", b, "gap before move %.4e" % float(gap0[b]), "indenter moved %.4e" % float(z0[b] - fem.ind[b, 3]), "gap after step %.4e" % float(gap1[b]),
                  "newton", si[b, 0], "pcg", si[b, 3], "stats", st[b], "finite", bool(torch.isfinite(fem.sim.x[b]).all()))
        nf = np.nonzero(~(si[:, 2].astype(int) & 1).astype(bool))[0][:2]
        for b in nf:
            print("  ok env", b, "gap before %.4e" % float(gap0[b]), "moved %.4e" % float(z0[b] - fem.ind[b, 3]), "newton", si[b, 0], "pcg", si[b, 3])
