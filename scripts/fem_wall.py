import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from tacex_amd.uipc.gelpad_scene import FemGelpad
B=int(sys.argv[1]) if len(sys.argv)>1 else 512
fem=FemGelpad(B,"cuda:0")
for i in range(6): fem.step(i)
torch.cuda.synchronize()
fem.ms_log=[]
t0=time.perf_counter()
for i in range(6,36): fem.step(i)
torch.cuda.synchronize()
wall=(time.perf_counter()-t0)/30*1e3
print("wall ms/step %.3f"%wall, "event mean %.3f"%np.mean(fem.ms_log), "n", len(fem.ms_log))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for i in range(36,46): fem.step(i)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
