"""Host-side cost of one C4 step (cProfile of the enqueuing thread, top entries by cumulative time)."""
import sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import torch, bench, gc
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev = "cuda:0"
fem = FemGelpad(512, dev, max_newton_iter=64, side_stream=True)
rig = bench.Rig(512, 240, 320, 1, False, dev, 1, 0, fem=fem)
for i in range(24):
    rig.step(i)
rig.finish(); torch.cuda.synchronize()
gc.collect(); gc.disable()
pr = cProfile.Profile()
pr.enable()
for i in range(24, 24 + 63):
    rig.step(i)
pr.disable()
rig.finish(); torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
