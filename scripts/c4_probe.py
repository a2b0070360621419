import sys, time; sys.path.insert(0,'.')
import torch
import bench
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev=torch.device("cuda:0")
B=512
fem=FemGelpad(B,"cuda:0")
rig=bench.Rig(B,240,320,1,False,dev,1,seed=7,gather="obs32",obs_dtype="u8",fem=fem)
fem.ms_log=[]
el=rig.timed(30,3)
ms=fem.ms_log[3:]
print("C4 ms/step %.3f  fem mean %.3f  n %d"%(el/30*1e3, sum(ms)/len(ms), len(ms)))
# split: sensors only
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(30):
    for s in rig.sensors: s.update(dt=0.01, force_recompute=True)
torch.cuda.synchronize(); print("sensor update only ms %.3f"%((time.perf_counter()-t0)/30*1e3))
t0=time.perf_counter()
for i in range(33,63): fem.step(i)
torch.cuda.synchronize(); print("fem only wall ms %.3f"%((time.perf_counter()-t0)/30*1e3))
