import sys, time, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim
from tacex_amd.utils.synthetic import synthetic_depth_maps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = (240, 320) if len(sys.argv) < 3 else (480, 640)
t = Taxim(device="cuda:0")
hm, ind = synthetic_depth_maps(B, H, W, seed=1, device="cuda")
out = torch.empty((B, H, W, 3), device="cuda")
for _ in range(3):
    t.render_direct(hm, False, ind, out=out)
torch.cuda.synchronize()
t.set_profiling((H, W), True)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    t.render_direct(hm, False, ind, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
prof = t.read_profile((H, W))
print(f"B={B} {H}x{W}: {dt*1e3:.3f} ms/step  {B/dt:.0f} frames/s")
for k, (ms, cnt) in prof.items():
    print(f"  {k:24s} {ms/max(cnt,1):8.4f} ms x{cnt}")
t.set_profiling((H, W), False)
t0 = time.perf_counter()
for _ in range(n):
    t.render_direct(hm, False, ind, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"no-profiling: {dt*1e3:.3f} ms/step  {B/dt:.0f} frames/s")
