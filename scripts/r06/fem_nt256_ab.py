"""the CU-resident Newton kernel on 256 threads (one wave per SIMD, 256 VGPRs + 74 AGPRs, ScratchSize 0) against the same
kernel on 512 threads (two waves per SIMD, 356 B/lane of scratch) - on a pad BOTH can hold (a thread owns a vertex: V <= 256).
TACEX_FEM_NT256 is read once per process: run this script twice.  Prints mean FEM ms per step over three periods of the breathing scene."""
import os, sys, time, torch
from tacex_amd.uipc.gelpad_scene import FemGelpad
mesh = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5,6,4").split(","))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
fem = FemGelpad(B, "cuda:0", max_newton_iter=64, mesh=mesh)
fem.ms_log = []
fem.info_sum = torch.zeros(4, dtype=torch.float64, device="cuda:0")
for i in range(24):
    fem.step(i)
base = fem.info_sum.clone()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(24, 24 + 63):
    fem.step(i)
fem.flush(); torch.cuda.synchronize(); el = time.perf_counter() - t0
ms = fem.ms_log[24:]
tot = (fem.info_sum - base).cpu().numpy()
print(f"NT256={os.environ.get('TACEX_FEM_NT256', '0')} mesh {mesh} V {fem.num_verts} T {fem.num_tets} B {B}: wall {el / 63 * 1e3:.3f} ms/step, fem_ms mean {sum(ms) / len(ms):.3f} "
      f"min {min(ms):.3f} max {max(ms):.3f}; newton/step {tot[0] / 63:.2f} pcg/newton {tot[3] / max(tot[0], 1e-9):.1f} iters_max {int(fem.iters_max)} resident {fem.sim.newton_kernel_resident}")
