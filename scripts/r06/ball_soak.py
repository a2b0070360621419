"""Soak of the bench's ball scene: 3000 steps (143 press-and-release periods) of 512 envs at the default tolerances; flags, the Newton cap,
finiteness and the ball's drift are checked every 50 steps (one read-back each)."""
import sys, time, torch
from tacex_amd.uipc.gelpad_scene import FemBallScene
B, N = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = FemBallScene(B, "cuda:0", max_newton_iter=64)
flags = torch.zeros(B, dtype=torch.int64, device="cuda:0")
nmax = torch.zeros((), dtype=torch.float64, device="cuda:0")
tot = torch.zeros(4, dtype=torch.float64, device="cuda:0")
t0 = time.perf_counter()
for i in range(N):
    sc.step(i)
    si = sc.sim.step_info
    flags |= si[:, 2].to(torch.int64)
    nmax = torch.maximum(nmax, si[:, 0].max())
    tot += si.mean(0)
    if i % 50 == 49:
        assert torch.isfinite(sc.sim.x).all() and torch.isfinite(sc.sim.q).all(), i
torch.cuda.synchronize()
el = time.perf_counter() - t0
q = sc.sim.q
print(f"{N} steps x {B} envs in {el:.1f} s ({el / N * 1e3:.2f} ms / step); flags OR-ed over all steps and envs: {int(flags.max())} (envs with any flag: {int((flags != 0).sum())}); "
      f"worst env of any step: {int(nmax)} Newton iterations; means per step: newton {float(tot[0]) / N:.2f}, pcg / newton {float(tot[3] / tot[0]):.1f}")
print(f"ball centre after the soak: x in [{float(q[:,0,0].min())*1e3:.3f}, {float(q[:,0,0].max())*1e3:.3f}] mm, y in [{float(q[:,0,1].min())*1e3:.3f}, {float(q[:,0,1].max())*1e3:.3f}] mm, "
      f"z - z0 in [{float((q[:,0,2]-0.0105).min())*1e6:.1f}, {float((q[:,0,2]-0.0105).max())*1e6:.1f}] um; max |A^T A - I| {float((q[:,1:].transpose(1,2) @ q[:,1:] - torch.eye(3, dtype=torch.float64, device='cuda:0')).abs().max()):.2e}")
assert int(flags.max()) == 0 and int(nmax) < 64
