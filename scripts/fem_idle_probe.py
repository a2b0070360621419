#!/usr/bin/env python3
"""Does a FEM step pay a runtime stall after the queue has been idle?  (sporadic ~50 ms steps in bench sweeps: profiles/r05_experiments.md section 10)
Steps the FemGelpad scene with host-side idle gaps of various lengths before a step and prints the wall time of that step."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tacex_amd.uipc.gelpad_scene import FemGelpad

fem = FemGelpad(512, "cuda:0", max_newton_iter=64)
for i in range(30):
    fem.step(i)
torch.cuda.synchronize()
i = 30
gaps = (0.0, 0.0, 0.05, 0.05, 0.2, 0.2, 1.0, 1.0, 3.0, 3.0, 0.0, 0.0) if len(sys.argv) < 2 else tuple([0.03, 0.06, 0.1, 0.15, 0.25] * int(sys.argv[1]))
slow = 0
for gap in gaps:
    time.sleep(gap)
    t0 = time.perf_counter()
    fem.step(i); i += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fem.step(i); i += 1
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    slow += (t1 - t0) > 0.01
    if len(sys.argv) < 2 or (t1 - t0) > 0.01:
        print(f"idle {gap:4.2f} s -> step {1e3 * (t1 - t0):8.2f} ms, next step {1e3 * (t2 - t1):8.2f} ms", flush=True)
print(f"{slow} of {len(gaps)} steps behind an idle gap took more than 10 ms")
