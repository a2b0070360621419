"""FEM micro-bench (BASELINE config 4 shard: 512 envs x ~2k-tet gelpad on one MI355X): element terms + Newton step."""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
from tacex_amd.uipc.uipc_object import gelpad_box_mesh

ap = argparse.ArgumentParser(); ap.add_argument("--envs", type=int, default=512); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
P, T = gelpad_box_mesh(8, 10, 4)  # 495 verts / 1920 tets
B = a.envs
sim = UipcSim(UipcSimCfg(device="cuda:0"), num_envs=B)
UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
sim.setup_sim()
top = np.where(P[:, 2] > P[:, 2].max() - 1e-9)[0]
aim = torch.from_numpy(P[top]).cuda()[None].repeat(B, 1, 1)
aim[:, :, 2] -= torch.linspace(0.0002, 0.0012, B, device="cuda", dtype=torch.float64)[:, None]
sim.set_constraints(top, aim)
sim.x_tilde = sim.x.clone()
def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
nt = B * len(T)
for name, fn, bytes_per_tet in [
    # bytes per env and tet that have to cross HBM: 96 B of vertex positions read + the outputs written.  The mesh constants of
    # SURVEY 8(d)'s 208 B figure (indices, Dm^-1, volume: 112 B) are per MESH, shared by all envs and L2-resident - counting them per env
    # (rounds 2-4) printed 9.6 "TB/s" for the energy+gradient call, above the 8 TB/s peak
    ("element_terms energy+grad+hess (assembled)", lambda: sim.element_terms(), 96 + 8 + 96 + 1152),
    ("element_terms energy+grad", lambda: sim.element_terms(hessian=False), 96 + 8 + 96),
    ("energy (line-search evaluation)", lambda: sim.energy(), 96),
    ("gradient (vertex gather)", lambda: sim.gradient(), 96 + 96),
]:
    dt = timeit(fn, a.iters)
    print(f"{name:46s} {dt*1e3:8.3f} ms  {nt/dt/1e9:7.2f} Gtet/s  {nt*bytes_per_tet/dt/1e9:8.1f} GB/s (per-env bytes: x read + outputs written; mesh constants in L2 excluded)")
x0 = sim.x.clone()
def newton():
    sim.x.copy_(x0); sim.newton_step()
dt = timeit(newton, 5)
st = sim.stats.cpu().numpy()
print(f"newton_step (PCG<= {sim.cfg.linear_system.max_iter}, tol {sim.cfg.linear_system.tol_rate}) {dt*1e3:8.3f} ms/iter for {B} envs; pcg iters mean {st[:,3].mean():.1f} max {st[:,3].max():.0f}; step mean {st[:,2].mean():.2f}")
sim.x.copy_(x0)
t0 = time.perf_counter(); sim.step(max_newton_iter=8); torch.cuda.synchronize()
print(f"UipcSim.step (8 Newton iters cap): {(time.perf_counter()-t0)*1e3:.2f} ms, iters {sim.last_newton_iters}")
sim.x.copy_(x0); sim.v.zero_()
def full_step():
    sim.x.copy_(x0); sim.v.zero_(); sim.step(max_newton_iter=8)
dt = timeit(full_step, 3)
print(f"UipcSim.step warm (8 Newton iters cap, device-side early exit): {dt*1e3:.2f} ms, iters {sim.last_newton_iters}")
# the C4 / C5 bench scene (back face attached, sphere indenter breathing in and out): ms per FEM step and solver statistics
import os
from tacex_amd.uipc.gelpad_scene import FemGelpad
fem = FemGelpad(B, "cuda:0", motion=os.environ.get("FEM_MOTION", "breathing"))
if os.environ.get("FEM_COARSE"):  # A/B: coarse grid cells per axis, e.g. FEM_COARSE=4,5,1
    fem.sim.cfg.linear_system.coarse_grid = tuple(int(v) for v in os.environ["FEM_COARSE"].split(","))
    fem.sim._precond_dirty = True
if os.environ.get("FEM_MESH"):  # A/B: the sphere as a rigid triangle mesh (icosphere with FEM_MESH subdivisions), indenter kind 4
    from tacex_amd.uipc.indenter_meshes import icosphere
    mv, mt = icosphere(fem.R, int(os.environ["FEM_MESH"]))
    fem.sim.set_indenter_mesh(mv, mt)
    fem.ind[:, 0] = 4.0
    fem.ind[:, 4] = 0.0
    print("mesh indenter:", len(mt), "triangles")
for i in range(6):
    fem.step(i)
torch.cuda.synchronize()
ms, nits, pcgs = [], [], []
for i in range(6, 30):
    fem.step(i)
    ms.append(fem.fem_ms_last())
    si = fem.sim.step_info.cpu().numpy()
    nits.append(si[:, 0].mean()); pcgs.append((si[:, 3] / np.maximum(si[:, 0], 1)).mean())
info = fem.sim.check_step(raise_on_penetration=False)
print(f"FemGelpad scene ({B} envs): {np.mean(ms):.3f} ms per step (median {np.median(ms):.3f}, min {np.min(ms):.3f}, max {np.max(ms):.3f}); Newton iterations per step mean "
      f"{np.mean(nits):.2f}; PCG iterations per Newton iteration mean {np.mean(pcgs):.1f}; flagged envs: penetration {len(info['penetrating_envs'])}, "
      f"line search {len(info['line_search_failed_envs'])}")
