#!/usr/bin/env python3
"""How often does a window of 21 FEM steps carry a stall?  (sporadic 20-50 ms steps in bench sweeps: profiles/r05_experiments.md section 13)
Runs N windows of 21 steps of (a) the FemGelpad scene alone, (b) the C4 rig of bench.py (sensor update + FEM on the side stream), a
synchronisation per window, and prints the distribution of the window times."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import bench
from tacex_amd.uipc.gelpad_scene import FemGelpad

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")


def windows(step, finish, label):
    import gc
    for i in range(45):
        step(i)
    finish(); torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t = []
    k = 45
    for w in range(N):
        t0 = time.perf_counter()
        for i in range(21):
            step(k); k += 1
        finish(); torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    gc.enable()
    t = np.array(t)
    med = float(np.median(t))
    slow = t > 1.3 * med
    print(f"{label}: {N} windows of 21 steps: median {med:.2f} ms, mean {t.mean():.2f}, max {t.max():.2f}; {int(slow.sum())} windows above 1.3 x median "
          f"({', '.join(f'#{i}: {t[i]:.1f}' for i in np.where(slow)[0][:12])}); mean / median = {t.mean() / med:.4f}", flush=True)


fem = FemGelpad(512, dev, max_newton_iter=bench.NEWTON_CAP)
windows(lambda i: fem.step(i), lambda: fem.flush(), "FemGelpad alone (512 envs, one stream)")
del fem
torch.cuda.empty_cache()
fem = FemGelpad(512, dev, max_newton_iter=bench.NEWTON_CAP, side_stream=True)
rig = bench.Rig(512, 240, 320, 1, False, dev, 1, seed=7, fem=fem)
fem.ms_log = []
windows(lambda i: rig.step(i), lambda: rig.finish(), "C4 rig (sensor update + FEM on the side stream)")
ms = np.array(fem.ms_log[45:])
print(f"  FEM part by hipEvents: median {np.median(ms):.3f} ms, max {ms.max():.3f}, steps above 5 ms: {int((ms > 5).sum())} of {len(ms)}")
