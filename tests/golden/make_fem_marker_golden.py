#!/usr/bin/env python3
"""Golden vectors for the FEM-driven marker set-up (container only; reads /root/reference at run time).

The reference module fem_based/sim/tactile_sensor_sapienipc_modified.py imports usdrt / IsaacLab and cannot be
imported, but `_gen_marker_grid` (VT:189-247), `_gen_marker_weight` (VT:249-329) and `in_hull` (geometry.py:86-100)
are plain NumPy / SciPy / sklearn.  Their source is extracted from the files where they lie with `ast`, executed with a
stub `self` (and a stub for the one usdrt call that returns the surface triangles) and the inputs / outputs are stored.
"""
import ast
import math
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
VT = Path("/root/reference/source/tacex/tacex/simulation_approaches/fem_based/sim/tactile_sensor_sapienipc_modified.py")
GEO = Path("/root/reference/source/tacex/tacex/simulation_approaches/fem_based/sim/utils/geometry.py")


def extract(path: Path, names):
    tree = ast.parse(path.read_text())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            out[node.name] = ast.get_source_segment(path.read_text(), node)
    return out


def main():
    from sklearn.neighbors import NearestNeighbors

    from tacex_amd.uipc.uipc_object import UipcObject, UipcObjectCfg, gelpad_box_mesh

    src = extract(VT, {"_gen_marker_grid", "_gen_marker_weight"})
    geo = extract(GEO, {"in_hull"})
    ns = {"np": np, "math": math, "NearestNeighbors": NearestNeighbors}
    exec(geo["in_hull"], ns)
    import textwrap

    for k, v in src.items():
        exec(textwrap.dedent(v), ns)

    # a gelpad-sized block seen from a camera 24 mm behind its back face, camera frame = (x, y, z forward)
    # (wider than the GelSight Mini pad so that every marker is well inside the top face: with markers next to the rim the
    # reference's 3-D nearest-face search picks side faces whose xy projection is singular and np.linalg.inv raises)
    P, T = gelpad_box_mesh(10, 8, 3, size=(0.030, 0.018, 0.0045))
    obj = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T))
    tri_g = obj.surface_triangles()
    ids = np.unique(tri_g.reshape(-1))
    remap = -np.ones(len(P), dtype=np.int64)
    remap[ids] = np.arange(len(ids))
    tri = remap[tri_g].astype(np.int32)
    surf_cam = P[ids].copy()
    surf_cam[:, 0] -= 0.011  # marker grid spans x in [-8, 16.5] mm, y in [-6, 6] mm around the camera axis
    surf_cam[:, 1] -= 0.009
    surf_cam[:, 2] += 0.024

    class _T:  # torch-like wrapper: `.cpu().numpy()`
        def __init__(self, a): self.a = a
        def cpu(self): return self
        def numpy(self): return self.a

    class _Attr:
        def __init__(self, a): self.a = a
        def Get(self): return self.a.reshape(-1)

    class _Mesh:
        def __init__(self, prim): self.prim = prim
        def GetFaceVertexIndicesAttr(self): return _Attr(tri)

    ns["usdrt"] = types.SimpleNamespace(UsdGeom=types.SimpleNamespace(Mesh=_Mesh))
    self = types.SimpleNamespace(
        marker_interval_range=(2.0625, 2.0625), marker_rotation_range=0.0, marker_translation_range=(0.0, 0.0),
        marker_pos_shift_range=(0.0, 0.0), init_surface_vertices_camera=_T(surf_cam.astype(np.float32)),
        gelpad_obj=types.SimpleNamespace(fabric_prim=None))
    np.random.seed(0)
    grid = ns["_gen_marker_grid"](self)
    idx, wgt = ns["_gen_marker_weight"](self, grid)
    # a second, randomised grid (rotation / translation / per-marker shift) with a fixed numpy seed
    self2 = types.SimpleNamespace(**{**self.__dict__, "marker_interval_range": (1.8, 2.2), "marker_rotation_range": 0.05,
                                     "marker_translation_range": (0.5, 0.4), "marker_pos_shift_range": (0.05, 0.05)})
    np.random.seed(123)
    grid2 = ns["_gen_marker_grid"](self2)
    np.savez_compressed(HERE / "fem_markers.npz", surf_cam=surf_cam.astype(np.float32), triangles=tri, grid=grid,
                        tri_idx=idx.astype(np.int32), weights=wgt, grid_random=grid2)
    print("grid", grid.shape, "on-surface markers", idx.shape, "grid_random", grid2.shape)


if __name__ == "__main__":
    main()
