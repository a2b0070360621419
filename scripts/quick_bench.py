"""Quick A/B bench of the Taxim render (no sensor boundary): per-stage hipEvent times. Usage:
   python scripts/quick_bench.py [B] [--hw 480x640] [--unfused] [--iters N]"""
import argparse, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim
from tacex_amd.utils.synthetic import synthetic_depth_maps
ap = argparse.ArgumentParser()
ap.add_argument("B", nargs="?", type=int, default=256)
ap.add_argument("--hw", default="240x320")
ap.add_argument("--unfused", action="store_true")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--no-profile", action="store_true")
a = ap.parse_args()
B = a.B
H, W = map(int, a.hw.split("x"))
t = Taxim(device="cuda:0")
hm, ind = synthetic_depth_maps(B, H, W, seed=1, device="cuda")
out = torch.empty((B, H, W, 3), device="cuda")
if a.unfused:
    t.set_fused_tail((H, W), False)
for _ in range(3):
    t.render_direct(hm, False, ind, out=out)
torch.cuda.synchronize()
n = a.iters
if not a.no_profile:
    t.set_profiling((H, W), True)
    for _ in range(n):
        t.render_direct(hm, False, ind, out=out)
    torch.cuda.synchronize()
    prof = t.read_profile((H, W))
    for k, (ms, cnt) in prof.items():
        if cnt:
            print(f"  {k:24s} {ms/max(cnt,1):8.4f} ms x{cnt}")
    t.set_profiling((H, W), False)
t0 = time.perf_counter()
for _ in range(n):
    t.render_direct(hm, False, ind, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B} {H}x{W} {'unfused' if a.unfused else 'fused'}: {dt*1e3:.3f} ms/step  {B/dt:.0f} frames/s")
