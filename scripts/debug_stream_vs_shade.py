"""debug: streaming tail vs stand-alone shade kernel on the 240x320 golden input; prints where they differ."""
import sys
from pathlib import Path
import numpy as np, torch
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R))
from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim
H, W = 240, 320
g = dict(np.load(R / "tests/golden" / f"taxim_{H}x{W}.npz"))
t = Taxim(calib_folder=str(R / "tacex_amd/assets/calib/gsmini_640x480"), backend="hip", device="cuda:0")
hm = torch.from_numpy(g["hm"]).cuda(); indent = torch.from_numpy(g["indent"]).cuda()
Z, M = t.deform(hm, indent)
rgb, idx = t.shade(Z, return_bins=True)
out = t.render_direct(hm, with_shadow=False, press_depth=indent).movedim(1, 3)
d = (out - rgb).abs().amax(-1)
bad = (d > 1e-6).nonzero()
print("n bad", bad.shape[0], "of", d.numel(), "max", d.max().item())
idx = idx.cpu().numpy()
for b, y, x in bad[:30].tolist():
    print(b, y, x, "d", d[b, y, x].item(), "bins", idx[b, y, x], "Z", Z[b, y, x].item())
if bad.shape[0]:
    ys = bad[:, 1].cpu().numpy(); xs = bad[:, 2].cpu().numpy()
    print("rows", np.unique(ys)[:40], "cols", np.unique(xs)[:40])
    print("im hist of bad", np.bincount(idx[bad[:,0].cpu(), bad[:,1].cpu(), bad[:,2].cpu(), 0]))
