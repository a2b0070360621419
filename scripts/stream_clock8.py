"""probe (library built with -DTACEX_STREAM_CLOCK8): eight-section cycle split of a shaded iteration of the streaming tail."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tacex_amd.calibration import CALIB_GELSIGHT_MINI
from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim
from tacex_amd.utils.synthetic import synthetic_depth_maps
B, H, W = 1024, 240, 320
t = Taxim(calib_folder=CALIB_GELSIGHT_MINI, backend="hip", device="cuda:0")
hm, ind = synthetic_depth_maps(B, H, W, seed=1, device="cuda:0")
out = torch.empty((B, H, W, 3), device="cuda:0")
for _ in range(3):
    t.render_direct(hm, False, ind, out=out)
torch.cuda.synchronize()
d = out.reshape(B, -1)[:, :36].reshape(B, 4, 9).double().cpu()
n = d[..., 8].sum()
names = ["row read-back", "bins + table/bg issue", "S/ring/contact stats", "levels", "last-level taps", "memory wait", "issue+poly+stores+obs", "next row scalars"]
tot = 0.0
for k, nm in enumerate(names):
    v = float(d[..., k].sum() / n); tot += v
    print(f"{nm:28s} {v:8.0f} cycles")
print(f"{'sum':28s} {tot:8.0f} cycles per shaded iteration; shaded rows per wave {float(d[..., 8].mean()):.1f}")
