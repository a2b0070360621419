"""Height-map source micro-bench: analytic indenters -> height map + frame min + indentation (one launch), B envs."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd import IndenterHeightMapSource
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 240, 320
src = IndenterHeightMapSource(B, "cuda:0")
g = torch.Generator().manual_seed(0)
u = torch.rand((B, 8), generator=g)
src.set(torch.randint(0, 4, (B,), generator=g).float().cuda(), (0.3 + 0.4 * u[:, 0]).cuda() * W, (0.3 + 0.4 * u[:, 1]).cuda() * H,
        (0.15 + 0.2 * u[:, 2]).cuda() * H, 3.14 * u[:, 3].cuda(), (0.2 + 1.3 * u[:, 4]).cuda(), 0.6 * W, 0.6 * H)
hm = torch.empty((B, H, W), device="cuda"); fmin = torch.empty(B, device="cuda"); ind = torch.empty(B, device="cuda")
for _ in range(3): src.fill(hm, fmin, ind, 0.0045, 0.024)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 50
for _ in range(n): src.fill(hm, fmin, ind, 0.0045, 0.024)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"indenter source B={B} {W}x{H}: {dt*1e6:.1f} us/launch, {B*H*W*4/dt/1e9:.0f} GB/s written, {B/dt/1e6:.2f} M frames/s")
