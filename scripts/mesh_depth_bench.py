"""Timing of the mesh depth source: 1024 envs x 320x240, icosphere indenters of growing triangle counts."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from oracle.mesh_depth_oracle import icosphere  # mesh generator only (test infrastructure; nothing is checked here)
from tacex_amd import MeshDepthSource

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for sub in (2, 3, 4):
    V, T = icosphere(0.004, sub)
    src = MeshDepthSource(V, T, B, "cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    src.pos[:, 0] = (torch.rand(B, device="cuda", generator=g) - 0.5) * 0.008
    src.pos[:, 1] = (torch.rand(B, device="cuda", generator=g) - 0.5) * 0.006
    src.pos[:, 2] = 0.029 + torch.rand(B, device="cuda", generator=g) * 0.003
    for _ in range(3):
        src()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        src()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    hit = torch.isfinite(src.depth).float().mean().item()
    print(f"{len(T):5d} triangles: {dt * 1e3:7.3f} ms per {B} depth images ({B / dt / 1e3:.0f} K images/s), {hit * 100:.1f} % of the pixels hit")
