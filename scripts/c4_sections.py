"""Where the C4 step's serial chain goes (one stream): scene driver | FEM step | optical | FEM-driven markers | pack, hipEvent means over a period."""
import sys
sys.path.insert(0, '/root/repo')
import torch, bench
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev = "cuda:0"
fem = FemGelpad(512, dev, max_newton_iter=64)
rig = bench.Rig(512, 240, 320, 1, False, dev, 1, 0, fem=fem)
s = rig.sensors[0]
for i in range(24):
    rig.step(i)
torch.cuda.synchronize()
names = ["scene driver", "tacex_fem_step", "optical (rows, levels, tail)", "FEM-driven markers", "pack"]
acc = [0.0] * len(names)
N = 42
import math
for i in range(24, 24 + N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    # scene driver (copied from FemGelpad._step)
    fem._pos[:, 0].fill_(fem._body_x + 0.0002 * math.sin(0.2 * i))
    fem.att.apply(fem.sim, fem._pos, fem._quat32)
    gap = fem.sim.contact_gaps().amin(1)
    c = 0.5 - 0.5 * math.cos(0.3 * i)
    target = torch.add(fem._z_rest_t, fem.depth, alpha=-c)
    z = fem.ind[:, 3]
    torch.maximum(target, torch.add(z, gap, alpha=-0.5), out=z)
    ev[1].record()
    fem.sim.step(max_newton_iter=64)
    ev[2].record()
    sim_m = s.marker_motion_simulator
    s.marker_motion_simulator = None
    s.update(dt=0.01, force_recompute=True)
    ev[3].record()
    s.marker_motion_simulator = sim_m
    res = sim_m.marker_motion_simulation()
    s._data.output["marker_motion"][:] = res
    ev[4].record()
    out = s._data.output
    rig.obs.pack_all({"rgb32_0": out["tactile_rgb_obs"], "indent_0": s.indentation_depth, "markers_0": out["marker_motion"]})
    rig.obs.gather_async()
    ev[5].record()
    torch.cuda.synchronize()
    for k in range(5):
        acc[k] += ev[k].elapsed_time(ev[k + 1])
for n, a in zip(names, acc):
    print(f"{n:32s} {a / N * 1e3:8.1f} us")
print("sum", sum(acc) / N, "ms")
