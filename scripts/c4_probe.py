import sys, time; sys.path.insert(0,'.')
import torch
import bench
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev=torch.device("cuda:0")
B=512
fem=FemGelpad(B,"cuda:0")
rig=bench.Rig(B,240,320,1,False,dev,1,seed=7,gather="obs32",obs_dtype="u8",fem=fem)
for i in range(11): rig.step(i)
rig.finish(); torch.cuda.synchronize()
def ev(): e=torch.cuda.Event(enable_timing=True); e.record(); return e
for i in range(11,14):
    t0=time.perf_counter(); e0=ev(); fem.step(i); e1=ev(); t1=time.perf_counter()
    for s in rig.sensors: s.update(dt=0.01, force_recompute=True)
    e2=ev(); t2=time.perf_counter(); torch.cuda.synchronize(); t3=time.perf_counter()
    print(i, "host fem %.2f host sensors %.2f sync wait %.2f | gpu fem %.2f gpu sensors %.2f"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,e0.elapsed_time(e1),e1.elapsed_time(e2)))
