"""`UipcObject` - a (batched) deformable tet-mesh object, counterpart of
source/tacex_uipc/tacex_uipc/objects/uipc_object.py:49-88,95-243,442-470 for the gelpad use case:
tet mesh + StableNeoHookean(youngs_poisson) + mass density, and - since round 6 - ONE free affine body per env
(`AffineBodyConstitutionCfg`, uipc_object.py:62-74: the ball of the reference's ball-rolling UIPC scene) given by its closed surface mesh.
USD prims, wildmeshing and render-mesh updates of the reference are out of scope."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from ..utils.configclass import configclass


def load_msh(path) -> tuple[np.ndarray, np.ndarray]:
    """Gmsh 2.2 ASCII tetrahedral mesh -> (points (V,3) float64, tets (T,4) int32)."""
    lines = Path(path).read_text().splitlines()
    i = lines.index("$Nodes")
    n = int(lines[i + 1])
    pts = np.array([[float(v) for v in lines[i + 2 + k].split()[1:4]] for k in range(n)], dtype=np.float64)
    j = lines.index("$Elements")
    m = int(lines[j + 1])
    tets = []
    for k in range(m):
        p = lines[j + 2 + k].split()
        if int(p[1]) == 4:
            ntags = int(p[2])
            tets.append([int(v) - 1 for v in p[3 + ntags:3 + ntags + 4]])
    if not tets:
        raise ValueError(f"{path}: no tetrahedra (element type 4) found")
    return pts, np.asarray(tets, dtype=np.int32)


def gelpad_box_mesh(nx=8, ny=10, nz=4, size=(0.02075, 0.02525, 0.0045)) -> tuple[np.ndarray, np.ndarray]:
    """Regular gelpad-sized block (gsmini_cfg.py:22-24: 20.75 x 25.25 x 4.5 mm), 6 tets per cell;
    the default 8 x 10 x 4 grid gives 495 vertices / 1920 tets (~2k tets, BASELINE.json config 4)."""
    xs, ys, zs = (np.linspace(0, size[d], n + 1) for d, n in enumerate((nx, ny, nz)))
    P = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)
    idx = lambda i, j, k: (i * (ny + 1) + j) * (nz + 1) + k
    tets = []
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                c = [idx(i + a, j + b, k + d) for a in (0, 1) for b in (0, 1) for d in (0, 1)]
                for p in ((1, 3), (3, 2), (2, 6), (6, 4), (4, 5), (5, 1)):
                    tets.append([c[0], c[p[0]], c[p[1]], c[7]])
    return P, np.asarray(tets, dtype=np.int32)


@configclass
class UipcObjectCfg:
    prim_path: str = "/World/envs/env_.*/gelpad"
    mesh_points: np.ndarray = None
    """(V,3) rest positions [m] (replaces the reference's TetMeshCfg / USD mesh look-up)."""
    mesh_tets: np.ndarray = None
    """(T,4) vertex indices."""
    mesh_tris: np.ndarray = None
    """(Nt,3) outward-oriented surface triangles - an affine body is given by its closed surface (the reference tet-meshes it with
    wildmeshing and libuipc takes the surface of that, uipc_object.py:168-192); derived from `mesh_tets` when absent."""
    mass_density: float = 1e3
    init_pos: tuple = (0.0, 0.0, 0.0)
    """World position of the object's origin in every env (`init_state.pos` of the reference's AssetBaseCfg; affine bodies only - the gelpad's
    mesh points are world coordinates)."""

    @configclass
    class AffineBodyConstitutionCfg:
        """uipc_object.py:62-74."""

        m_kappa: float = 100.0
        """Stiffness of the body in [MPa] (100 MPa = hard rubber): weight of the orthogonality energy kappa * vol * |A^T A - I|^2"""
        kinematic: bool = False
        """True: the body's degrees of freedom are fixed within a step (uipc_object.py:463-466 `is_fixed`): the caller moves `UipcSim.q` between
        steps (the velocity `UipcSim.qv` then only feeds the inertia term, which a fixed body does not use)."""

    @configclass
    class StableNeoHookeanCfg:
        youngs_modulus: float = 0.01
        """in [MPa] (uipc_object.py:76-80)"""
        poisson_rate: float = 0.49

    constitution_cfg: StableNeoHookeanCfg = StableNeoHookeanCfg()
    attachment_cfg: object = None


class UipcObject:
    """Holds the mesh + material of one deformable object replicated over all envs of a UipcSim."""

    def __init__(self, cfg: UipcObjectCfg, uipc_sim=None):
        self.is_affine_body = isinstance(cfg.constitution_cfg, UipcObjectCfg.AffineBodyConstitutionCfg)
        if cfg.mesh_points is None or (cfg.mesh_tets is None and not (self.is_affine_body and cfg.mesh_tris is not None)):
            raise ValueError("UipcObjectCfg.mesh_points / mesh_tets are required (an affine body may give mesh_tris instead of mesh_tets)")
        self.cfg = cfg
        self.points = np.ascontiguousarray(cfg.mesh_points, dtype=np.float64)
        self.tets = np.ascontiguousarray(cfg.mesh_tets if cfg.mesh_tets is not None else np.zeros((0, 4)), dtype=np.int32)
        if self.points.ndim != 2 or self.points.shape[1] != 3 or self.tets.ndim != 2 or self.tets.shape[1] != 4:
            raise ValueError("mesh_points must be (V,3) and mesh_tets (T,4)")
        if self.is_affine_body:
            self.tris = np.ascontiguousarray(cfg.mesh_tris if cfg.mesh_tris is not None else self.surface_triangles(), dtype=np.int32)
            a, b, c = (self.points[self.tris[:, k]] for k in range(3))
            if np.einsum("ij,ij->i", a, np.cross(b, c)).sum() < 0.0:  # (boundary faces of negatively oriented tets look inward)
                self.tris = np.ascontiguousarray(self.tris[:, [0, 2, 1]])
        self._uipc_sim = uipc_sim
        if uipc_sim is not None:
            uipc_sim.uipc_objects.append(self)

    @property
    def num_verts(self) -> int:
        return self.points.shape[0]

    @property
    def num_tets(self) -> int:
        return self.tets.shape[0]

    def surface_vertex_areas(self) -> np.ndarray:
        """(V,) a third of the rest area of the surface triangles around each vertex (0 for interior vertices): the weight of a
        vertex in the contact barrier."""
        tri = self.surface_triangles()
        p = self.points
        a = 0.5 * np.linalg.norm(np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]]), axis=1)
        w = np.zeros(self.num_verts)
        np.add.at(w, tri.reshape(-1), np.repeat(a / 3.0, 3))
        return w

    def surface_triangles(self) -> np.ndarray:
        """Boundary faces (each appears in exactly one tet), oriented outward for positively oriented tets."""
        faces = {}
        for t in self.tets:
            for f in ((t[0], t[2], t[1]), (t[0], t[1], t[3]), (t[1], t[2], t[3]), (t[0], t[3], t[2])):
                key = tuple(sorted(f))
                faces[key] = None if key in faces else f
        return np.asarray([f for f in faces.values() if f is not None], dtype=np.int32)

    # -- uipc_object.py:280-370 ------------------------------------------------------------------------------------------------
    def reset(self, env_ids=None):
        """Rest positions and zero velocity for the envs in `env_ids` (None: all).  (A TODO stub in the reference, uipc_object.py:280-286:
        its scenes hold one env; with hundreds of envs per GPU an RL task resets single envs every few hundred steps.)"""
        if self._uipc_sim is None or getattr(self._uipc_sim, "_handle", None) is None:
            raise RuntimeError("UipcObject.reset needs a UipcSim that was set up (setup_sim)")
        self._uipc_sim.reset(env_ids)

    def write_vertex_positions_to_sim(self, vertex_positions, env_ids=None):
        """`world.write_vertex_pos_to_sim` of the reference's libuipc fork (uipc_object.py:318-370): vertex positions
        (len(env_ids), V, 3) in the simulation frame for the listed envs (None: all), velocities zeroed."""
        if self._uipc_sim is None or getattr(self._uipc_sim, "_handle", None) is None:
            raise RuntimeError("UipcObject.write_vertex_positions_to_sim needs a UipcSim that was set up (setup_sim)")
        self._uipc_sim.reset(env_ids, vertex_positions=vertex_positions)
