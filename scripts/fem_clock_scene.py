"""Section clock (group 3: phases of a Newton iteration) on the C4 gelpad scene: TACEX_LIB_TAG=fc3 TACEX_LIB_FROZEN=1 python scripts/fem_clock_scene.py"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tacex_amd.uipc.gelpad_scene import FemGelpad
fem = FemGelpad(512, "cuda:0", max_newton_iter=64)
names = ["gradient + contact (+ lag, friction)", "block assembly + chain factor", "PCG", "line search + update"]
for i in range(24):
    fem.step(i)
    torch.cuda.synchronize()
    st = fem.sim.stats.cpu().numpy()   # cycles summed over the env's Newton iterations of this step
    si = fem.sim.step_info.cpu().numpy()
    if i in (5, 6, 7, 16, 17, 18):
        print(f"step {i}: newton mean {si[:,0].mean():.2f} pcg mean {si[:,3].mean():.1f} | " + ", ".join(f"{n} {st[:, k].mean() / 1e3:.1f} K" for k, n in enumerate(names)) + f" | sum {st.sum(1).mean() / 1e3:.1f} K cycles = {st.sum(1).mean() / 2.4e3:.1f} us at 2.4 GHz", flush=True)
