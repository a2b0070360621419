"""probe (library built with -DTACEX_STREAM_CLOCK): per-iteration cycle split of the streaming tail around its one memory wait."""
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
d = out.reshape(B, -1)[:, :16].reshape(B, 4, 4).double().cpu()
n = d[..., 3].sum()
print("per shaded row (cycles, s_memtime @100MHz?): before-wait %.0f  wait %.0f  after-wait %.0f   rows/wave %.1f" % (
    d[..., 0].sum() / n, d[..., 1].sum() / n, d[..., 2].sum() / n, d[..., 3].mean()))
print("per-slot means:", (d[..., :3].sum(0) / d[..., 3:4].sum(0)).tolist())
