"""A pad beyond the CU-resident kernel's reach (11 x 13 x 5 = 715 vertices, 2880 tets) on the streaming Newton kernel: FEM ms per step of the
breathing scene, 512 envs.  TACEX_FEM_STREAM_LDS=0/1 (x, p, accumulators in LDS) is read once per process."""
import os, sys, time, torch
from tacex_amd.uipc.gelpad_scene import FemGelpad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fem = FemGelpad(B, "cuda:0", max_newton_iter=64, mesh=(10, 12, 4))
fem.ms_log = []
fem.info_sum = torch.zeros(4, dtype=torch.float64, device="cuda:0")
for i in range(24):
    fem.step(i)
base = fem.info_sum.clone()
for i in range(24, 24 + 42):
    fem.step(i)
fem.flush(); torch.cuda.synchronize()
ms = fem.ms_log[24:]
tot = (fem.info_sum - base).cpu().numpy()
print(f"STREAM_LDS={os.environ.get('TACEX_FEM_STREAM_LDS', '1')} V {fem.num_verts} T {fem.num_tets} B {B}: fem_ms mean {sum(ms) / len(ms):.3f} min {min(ms):.3f} max {max(ms):.3f}; "
      f"newton/step {tot[0] / 42:.2f} pcg/newton {tot[3] / max(tot[0], 1e-9):.1f} iters_max {int(fem.iters_max)} resident {fem.sim.newton_kernel_resident}")
