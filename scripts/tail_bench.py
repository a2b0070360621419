"""Per-stage hipEvent timings of the Taxim render (library-side events) for tail-kernel variants.
   usage: python scripts/tail_bench.py [B] [mode]   mode: 1 = streaming tail (default), 2 = LDS-tiled tail"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tacex_amd.calibration import CALIB_GELSIGHT_MINI
from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim
from tacex_amd.utils.synthetic import synthetic_depth_maps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (240, 320)
t = Taxim(calib_folder=CALIB_GELSIGHT_MINI, backend="hip", device="cuda:0")
hm, ind = synthetic_depth_maps(B, H, W, seed=1, device="cuda:0")
out = torch.empty((B, H, W, 3), device="cuda:0")
obs = torch.empty((B, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
t.set_fused_tail((H, W), mode)
for with_obs in (False, True):
    for _ in range(3):
        t.render_direct(hm, False, ind, out=out, obs_out=obs if with_obs else None)
    t.set_profiling((H, W), True)
    for _ in range(20):
        t.render_direct(hm, False, ind, out=out, obs_out=obs if with_obs else None)
    torch.cuda.synchronize()
    prof = t.read_profile((H, W))
    t.set_profiling((H, W), False)
    tot = 0.0
    parts = []
    for k, (ms, n) in prof.items():
        if n:
            parts.append(f"{k}={ms / n * 1e3:.1f}us x{n // 20}")
            tot += ms / 20
    print(f"B={B} {W}x{H} mode={mode} obs={with_obs}: total {tot * 1e3:.1f} us/render ({B / tot / 1e3 * 1e3:.0f} frames/s)  " + "  ".join(parts), flush=True)
