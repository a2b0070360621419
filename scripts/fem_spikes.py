"""Wall time of every FemGelpad step over a long run: do the one-time ~50 ms runtime stalls of the first long Newton launches recur?"""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from tacex_amd.uipc.gelpad_scene import FemGelpad
fem = FemGelpad(512, "cuda:0")
ts = []
for i in range(400):
    t0 = time.perf_counter(); fem.step(i); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("steps > 30 ms:", [(int(i), round(float(t), 1)) for i, t in enumerate(ts) if t > 30])
print("mean %.2f ms, p50 %.2f, p99 %.2f, max %.2f" % (ts.mean(), np.median(ts), np.quantile(ts, 0.99), ts.max()))
