"""Is the C4 step host-bound?  Enqueue time of the Python loop against the time until the GPU is done."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, bench, gc
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev = "cuda:0"
for side in (True, False):
    fem = FemGelpad(512, dev, max_newton_iter=64, side_stream=side)
    rig = bench.Rig(512, 240, 320, 1, False, dev, 1, 0, fem=fem)
    for i in range(24):
        rig.step(i)
    rig.finish(); torch.cuda.synchronize()
    gc.collect(); gc.disable()
    N = 63
    t0 = time.perf_counter()
    for i in range(24, 24 + N):
        rig.step(i)
    t1 = time.perf_counter()
    rig.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    gc.enable()
    print(f"side_stream={side}: enqueue {(t1 - t0) / N * 1e3:.3f} ms per step, until done {(t2 - t0) / N * 1e3:.3f} ms per step", flush=True)
    del rig, fem
    torch.cuda.empty_cache()
