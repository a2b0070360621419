"""`UipcSim` - batched gelpad FEM stepping, counterpart of source/tacex_uipc/tacex_uipc/sim/uipc_sim.py:32-131,134-312.

The reference wraps libuipc (one scene, num_envs = 1 in practice, docs/source/showcases/ball_rolling.md:23) and
calls `world.advance(); world.retrieve()` (uipc_sim.py:250-252).  Here the Newton loop of that advance - element
gradient/Hessian work, PCG and line search - runs in HIP for `num_envs` independent copies of the gelpad.
IPC contact against prescribed indenters (analytic / one rigid mesh) since rounds 2-5; since round 6 also the reference's own UIPC
scene - ONE free affine-body ball per env on the ground plane under the gelpad, point-triangle pairs in both directions
(`UipcObjectCfg.AffineBodyConstitutionCfg`, `UipcSimCfg.ground_height`; csrc/fem_ball.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..utils.configclass import configclass


@configclass
class UipcSimCfg:
    """Field names follow uipc_sim.py:32-131 (entries that only configure out-of-scope libuipc parts are kept
    for drop-in construction and ignored)."""

    device: str = "cuda"
    dt: float = 0.01
    sanity_check_enable: bool = True
    sanity_check_mode: str = "quiet"
    workspace: str = ""
    logger_level: str = "Error"
    gravity: tuple = (0.0, 0.0, -9.8)
    ground_height: float = 0.0
    ground_normal: tuple = (0.0, 0.0, 1.0)

    @configclass
    class Newton:
        max_iter: int = 1024
        use_adaptive_tol: bool = False
        velocity_tol: float = 0.05
        """converged when max |dx| / dt <= velocity_tol [m/s]"""
        ccd_tol: float = 1.0
        transrate_tol: float = 0.1

    newton: Newton = Newton()

    @configclass
    class LinearSystem:
        solver: str = "linear_pcg"
        tol_rate: float = 1e-3
        """The PCG of a Newton iteration stops when r^T M^-1 r <= tol_rate x its value for the right-hand side - libuipc's test (`LinearPCG::pcg`:
        `abs(rz_new) <= global_tol_rate * rz0`), relative on r.z itself: sqrt(tol_rate) = 0.032 on the M^-1 norm of the residual."""
        max_iter: int = 1024
        """PCG iteration cap (not in the reference cfg, uipc_sim.py:86-90: libuipc stops on `tol_rate`); only guards against stagnation."""
        deterministic: bool = False
        """True: tet -> vertex sums in a fixed order (bit-identical runs); False: LDS atomics (`tacex_fem_set_deterministic`): two runs
        agree to round-off, a PCG iteration is ~25 % cheaper.  Not in the reference cfg."""
        coarse_grid: tuple | str | None = "auto"
        """Coarse grid (cells per axis) of the additive coarse correction beside the 3x3 block Jacobi (`coarse_space.py`):
        "auto" = 3 cells along the longest extent of the mesh, proportionally fewer along the others (2 x 3 x 1 = 24 nodes for the
        gelpad); None = block Jacobi alone (120-330 PCG iterations on the gelpad).  Not in the reference cfg.
        A grid NESTED in a structured mesh (nodes on mesh vertices: (4, 5, 1) on the 8 x 10 x 4 pad) is the better coarse space where the
        right-hand side is smooth - the affine-body scene, `FemBallScene`: 12.9 -> 9.6 PCG iterations per Newton iteration - but NOT a
        safe choice for the prescribed-indenter scenes on the CU-resident kernel: on C4's breathing press one env of 512 met negative
        curvature, lost its coarse correction and ran into the Newton cap with a penetrated state (flags 13;
        profiles/r06_experiments.md section 10, scripts/r06/c4_grid451_probe.py).  `check_step()` reports such envs."""
        vertex_chains: list | str | None = "auto"
        """Vertex chains of the block part of the preconditioner (`tacex_fem_set_chains`): "auto" = the columns of vertices through the
        mesh's thin direction (`coarse_space.build_vertex_chains`; an unstructured mesh yields none), a list of vertex-id lists, or
        None = one 3x3 block per vertex.  With the coarse correction the chains take the pad's free motion from 67 to 27 PCG
        iterations per Newton iteration.  Not in the reference cfg."""

    linear_system: LinearSystem = LinearSystem()

    @configclass
    class LineSearch:
        max_iter: int = 8
        report_energy: bool = False
        refine: int = 4
        """Scenes with an affine body (`tacex_fem_set_line_search_refine`): bisections between the accepted and the last rejected step of a
        backtracking search that had to cut the step - the largest step that still does not increase the potential takes the pair that cut it
        INSIDE the barrier zone, where the next iteration's Hessian sees it (profiles/r06_experiments.md sections 15-16).  0 = plain halving.
        Not in the reference cfg (uipc_sim.py:96-101)."""

    line_search: LineSearch = LineSearch()

    @configclass
    class Contact:
        """uipc_sim.py:103-124.  IPC contact of the gelpad surface against ONE analytic indenter per env
        (`UipcSim.set_contact_indenters`): barrier + CCD step bound, and lagged Coulomb friction (`enable_friction`,
        `default_friction_ratio`, `eps_velocity`) relative to the indenter's own motion between steps.  Mesh-mesh contact is not
        implemented."""

        enable: bool = True
        enable_friction: bool = True
        default_friction_ratio: float = 0.5
        default_contact_resistance: float = 10.0
        """in [GPa]; the barrier stiffness is resistance * d_hat [J/m^2] per unit of surface area"""
        constitution: str = "ipc"
        d_hat: float = 0.001
        eps_velocity: float = 0.01
        friction_lag: str = "ipc"
        """Where the friction lag (normal force, normal; frozen per time step) comes from (`tacex_fem_set_friction_lag`).  "ipc" (default
        since round 6) = the previous configuration, Li et al. 2020 section 5.4 to the letter: the step's end state is a stationary point
        of IPC's plain incremental potential (tests/test_fem_physics_gpu.py), and on the gelpad scenes `north_star` names it is also the
        faster rule (C4: 408.9 K against 384.5 K frames/s, BENCH_r05).  "capped" (opt-in) = start positions against the indenter's NEW
        position, the force capped by the contact reaction there: agrees with "ipc" where the previous step converged tightly and the
        indenter approaches, takes the smaller, already relaxed force when it retreats (end state up to ~150 um off IPC's on such steps),
        and stays bounded at the reference's default Newton tolerance on SLENDER bodies, where the previous configuration is far from
        balance (simple_axle at the defaults: 302 K env steps/s against 8.5 K with "ipc", profiles/r05_experiments.md) - the setting for
        those.  Not in the reference cfg (libuipc's rule is not in the reference)."""
        edge_edge: bool = True
        """Edge-edge pairs between the gelpad's surface and an affine body's (`tacex_fem_set_edge_edge`): IPC's contact set is point-triangle
        AND edge-edge pairs (Li et al. 2020); False leaves the point-triangle pairs of both directions alone (the first cut of this scene -
        between a pad of ~1.5 mm triangles and a ball of 2.7 mm edges at d_hat 0.5 mm the point-triangle pairs already keep the surfaces
        apart; the switch is there to compare).  Not a field of the reference's cfg: libuipc has no such switch."""
        follow_indenter: bool = True
        """Contact-following start of a step's Newton loop (`tacex_fem_set_contact_following`): vertices inside the barrier zone start
        the iteration displaced with their indenter.  An initial guess only (same minimiser); not in the reference cfg - libuipc starts
        from the current positions, which makes a retreating indenter cost 4-30 Newton iterations instead of 2-3."""

    contact: Contact = Contact()
    collision_detection_method: str = "linear_bvh"
    diff_sim: bool = False


class UipcSim:
    cfg: UipcSimCfg

    def __init__(self, cfg: UipcSimCfg, num_envs: int = 1):
        self.cfg = cfg
        self.num_envs = int(num_envs)
        self.uipc_objects = []
        self._handle = None
        self._lib = None
        self.device = torch.device(cfg.device)
        if self.device.type != "cuda":
            raise _lib.TacexHipError("UipcSim needs an AMD GPU device (no CPU fallback)")
        self._dev_index = self.device.index if self.device.index is not None else 0

    # -- uipc_sim.py:228-248 -------------------------------------------------------------------------------
    def setup_sim(self, constraint_strength_ratio: float | None = None):
        soft = [o for o in self.uipc_objects if not getattr(o, "is_affine_body", False)]
        body = [o for o in self.uipc_objects if getattr(o, "is_affine_body", False)]
        if len(soft) != 1 or len(body) > 1:
            raise RuntimeError("this build steps exactly one deformable object (the gelpad) and at most one free affine body per environment")
        obj = soft[0]
        _lib.require_gpu(self._dev_index)
        lib = _lib.load_library()
        p = _lib.FemParams()
        p.num_verts, p.num_tets = obj.num_verts, obj.num_tets
        p.rest_positions = obj.points.ctypes.data_as(C.POINTER(C.c_double))
        p.tets = obj.tets.ctypes.data_as(_lib.c_int32_p)
        p.youngs = obj.cfg.constitution_cfg.youngs_modulus * 1e6  # MPa -> Pa (uipc_object.py:452)
        p.poisson = obj.cfg.constitution_cfg.poisson_rate
        p.density = obj.cfg.mass_density
        p.dt = self.cfg.dt
        for k in range(3):
            p.gravity[k] = self.cfg.gravity[k]
        if constraint_strength_ratio is None:
            ac = obj.cfg.attachment_cfg
            constraint_strength_ratio = ac.constraint_strength_ratio if ac is not None else 100.0
        p.constraint_strength_ratio = constraint_strength_ratio
        self._strength = float(constraint_strength_ratio)
        self._precond_dirty = True
        h = C.c_void_p()
        _lib.check(lib.tacex_fem_create(self._dev_index, C.byref(p), C.byref(h)), "tacex_fem_create")
        self._handle, self._lib, self._obj = h, lib, obj
        _lib.check(lib.tacex_fem_set_deterministic(h, 1 if self.cfg.linear_system.deterministic else 0), "tacex_fem_set_deterministic")
        B, V = self.num_envs, obj.num_verts
        dev = self.device
        self.x = torch.from_numpy(obj.points).to(dev)[None].repeat(B, 1, 1).contiguous()  # (B,V,3) float64
        self.v = torch.zeros_like(self.x)
        self.x_tilde = torch.empty_like(self.x)
        self.is_constrained = torch.zeros((B, V), dtype=torch.uint8, device=dev)
        self.aim_position = self.x.clone()
        self.stats = torch.zeros((B, 4), dtype=torch.float64, device=dev)
        self._ws = torch.empty(lib.tacex_fem_workspace_bytes(h, B), dtype=torch.uint8, device=dev)
        self._g = torch.tensor(self.cfg.gravity, dtype=torch.float64, device=dev)
        self._body = None
        if body:
            self._setup_affine_body(body[0])

    # -- the reference's UIPC scene: a free affine body + the ground (ball_rolling_uipc.py:71-92, uipc_sim.py:192-201) ----------------
    def _setup_affine_body(self, body):
        """One free affine body per env (`UipcObjectCfg.AffineBodyConstitutionCfg`): q (num_envs,4,3) = (p, c_1, c_2, c_3), c_k the columns
        of A, a surface point is p + sum_k X_k c_k.  Contact: the ground half-space z >= cfg.ground_height and every point-triangle and
        edge-edge pair between the gelpad's surface and the body's closer than cfg.contact.d_hat (`tacex_fem_set_affine_body`)."""
        if tuple(float(v) for v in self.cfg.ground_normal) != (0.0, 0.0, 1.0):
            raise NotImplementedError("the ground of a scene with an affine body is the half-space z >= ground_height (ground_normal (0, 0, 1))")
        obj, B, dev = self._obj, self.num_envs, self.device
        c = self.cfg.contact
        if not c.enable:
            raise NotImplementedError("a scene with an affine body needs contact (UipcSimCfg.contact.enable): nothing else holds the free body")
        if self.cfg.linear_system.deterministic:
            raise NotImplementedError("linear_system.deterministic is not available with an affine body (csrc/fem_ball.h adds its tet / pair rows with LDS atomics)")
        area = np.ascontiguousarray(obj.surface_vertex_areas(), np.float64)
        ptri = np.ascontiguousarray(obj.surface_triangles(), np.int32)
        verts = np.ascontiguousarray(body.points, np.float64)
        tris = np.ascontiguousarray(body.tris, np.int32)
        d_hat = float(c.d_hat)
        _lib.check(self._lib.tacex_fem_set_affine_body(
            self._handle, len(verts), verts.ctypes.data, len(tris), tris.ctypes.data, float(body.cfg.mass_density),
            float(body.cfg.constitution_cfg.m_kappa) * 1e6, area.ctypes.data, len(ptri), ptri.ctypes.data, d_hat,
            float(c.default_contact_resistance) * 1e9 * d_hat, float(self.cfg.ground_height), 1 if c.enable else 0,
            1 if body.cfg.constitution_cfg.kinematic else 0), "tacex_fem_set_affine_body")
        _lib.check(self._lib.tacex_fem_set_edge_edge(self._handle, 1 if getattr(c, "edge_edge", True) else 0), "tacex_fem_set_edge_edge")
        _lib.check(self._lib.tacex_fem_set_line_search_refine(self._handle, int(getattr(self.cfg.line_search, "refine", 4))), "tacex_fem_set_line_search_refine")
        # one default contact model for every pair of surfaces (US:192-201): friction ratio / eps_velocity of the cfg act on the pairs and the ground
        _lib.check(self._lib.tacex_fem_set_friction(self._handle, float(c.default_friction_ratio) if c.enable_friction else 0.0,
                                                    float(c.eps_velocity)), "tacex_fem_set_friction")
        self._body = body
        q0 = np.concatenate([np.asarray(body.cfg.init_pos, np.float64)[None], np.eye(3)], 0)
        self._q0 = torch.from_numpy(q0).to(dev)
        self.q = self._q0[None].repeat(B, 1, 1).contiguous()  # (B,4,3) float64
        self.qv = torch.zeros_like(self.q)
        self._body_Y = torch.from_numpy(np.concatenate([np.ones((len(verts), 1)), verts], 1)).to(dev)
        self._ball_ws = torch.empty(self._lib.tacex_fem_ball_workspace_bytes(self._handle, B), dtype=torch.uint8, device=dev)

    def body_points(self, q=None) -> torch.Tensor:
        """(num_envs, nv, 3) world positions of the affine body's surface vertices."""
        q = self.q if q is None else q
        return torch.einsum("va,bai->bvi", self._body_Y, q)

    def ball_terms(self, x=None, q=None, x_tilde=None, q_tilde=None, constrained=True, x_prev=None, q_prev=None):
        """Energy (num_envs,) and gradient (num_envs, V + 4, 3) of the step's incremental potential at (x, q) (`tacex_fem_ball_terms`).
        x_prev / q_prev: the state friction slides relative to (None: friction off in this evaluation)."""
        x = self.x if x is None else x
        q = self.q if q is None else q
        xt = x if x_tilde is None else x_tilde
        qt = q if q_tilde is None else q_tilde
        B, V = x.shape[0], x.shape[1]
        E = torch.empty((B,), dtype=torch.float64, device=self.device)
        g = torch.empty((B, V + 4, 3), dtype=torch.float64, device=self.device)
        si = torch.zeros((B, 4), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_ball_terms(self._handle, _lib.ptr(x), _lib.ptr(xt), _lib.ptr(q), _lib.ptr(qt),
                                                _lib.ptr(self.is_constrained) if constrained else 0, _lib.ptr(self.aim_position) if constrained else 0,
                                                _lib.ptr(x_prev) if x_prev is not None else 0, _lib.ptr(q_prev) if q_prev is not None else 0,
                                                _lib.ptr(E), _lib.ptr(g), _lib.ptr(si), _lib.ptr(self._ball_ws), B, self._stream())
        _lib.check(rc, "tacex_fem_ball_terms")
        return E, g, si

    def __del__(self):
        try:
            if self._handle:
                self._lib.tacex_fem_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _stream(self):
        return _lib.current_stream_handle(self.device)

    # -- IPC contact against analytic indenters (first slice) ------------------------------------------------------
    def set_contact_indenters(self, indenters: torch.Tensor | None):
        """One analytic indenter per env the gelpad surface may not penetrate: (num_envs, 8) float64
        [kind, cx, cy, cz, radius, nx, ny, nz], kind 0 none, 1 sphere (centre c, radius), 2 half-space (unit normal n through c,
        the solid side is n.(x - c) < 0), 3 capsule (centre c, radius, (nx, ny, nz) = HALF the axis vector: a lying pin, a
        finger).  None disables contact.  Needs `cfg.contact.enable`.

        Contract: between two `step()` calls an indenter may approach the pad by LESS than the current gap (a rigid-body
        integrator with CCD guarantees that; `contact_gaps()` gives the gap).  A vertex found at or beyond the surface has infinite
        barrier energy and no gradient: the step still runs, the env is flagged in `step_info[:, 2]` and `check_step()` raises.

        Moving the indenter: mutate `sim.contact_indenters` in place, or call this again with a new tensor every step - both are
        seen by friction as the indenter's displacement since the previous `step()` (the library keeps the previous positions
        across this call; disabling contact resets them).  Only the TRANSLATION (cx, cy, cz) counts: a rotation of a capsule /
        mesh indenter between two steps does not drag the pad (`tacex_fem_set_friction`, include/tacex_hip.h)."""
        if indenters is None:
            _lib.check(self._lib.tacex_fem_set_contact(self._handle, 0, 0.0, 0.0, 0), "tacex_fem_set_contact")
            self.contact_indenters = None
            return
        if not self.cfg.contact.enable:
            raise RuntimeError("UipcSimCfg.contact.enable is False")
        if getattr(self, "_body", None) is not None:
            raise NotImplementedError("a scene with an affine body takes its contact from the body, the ground and the gelpad (pairs); prescribed "
                                      "indenters are the other kind of scene (or make the body kinematic: AffineBodyConstitutionCfg.kinematic)")
        ind = indenters.to(self.device, torch.float64).reshape(self.num_envs, 8).contiguous()
        area = None
        if not getattr(self, "_contact_area_set", False):
            area = np.ascontiguousarray(self._obj.surface_vertex_areas(), dtype=np.float64)
            self._contact_area_set = True
        d_hat = float(self.cfg.contact.d_hat)
        stiffness = float(self.cfg.contact.default_contact_resistance) * 1e9 * d_hat
        _lib.check(self._lib.tacex_fem_set_contact(self._handle, area.ctypes.data if area is not None else 0, d_hat, stiffness,
                                                   _lib.ptr(ind)), "tacex_fem_set_contact")
        c = self.cfg.contact
        _lib.check(self._lib.tacex_fem_set_friction(self._handle, float(c.default_friction_ratio) if c.enable_friction else 0.0,
                                                    float(c.eps_velocity)), "tacex_fem_set_friction")
        _lib.check(self._lib.tacex_fem_set_contact_following(self._handle, 1 if getattr(c, "follow_indenter", True) else 0),
                   "tacex_fem_set_contact_following")
        lag = getattr(c, "friction_lag", "ipc")
        if lag not in ("ipc", "capped"):
            raise ValueError(f"UipcSimCfg.contact.friction_lag must be 'ipc' or 'capped', got {lag!r}")
        _lib.check(self._lib.tacex_fem_set_friction_lag(self._handle, 1 if lag == "ipc" else 0), "tacex_fem_set_friction_lag")
        self.contact_indenters = ind  # keeps the device buffer alive: the kernels read it on every later call

    def set_indenter_mesh(self, vertices, triangles):
        """Rigid triangle mesh for indenter kind 4 (one mesh per scene, shared by the envs; every env places it with its indenter row):
        vertices (Nv,3) in the mesh's own frame, triangles (Nt,3).  The barrier acts between every surface vertex of the gelpad and
        its nearest triangle (`tacex_fem_set_indenter_mesh`).  None removes it."""
        if vertices is None:
            _lib.check(self._lib.tacex_fem_set_indenter_mesh(self._handle, 0, 0, 0, 0), "tacex_fem_set_indenter_mesh")
            self.indenter_mesh = None
            return
        v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
        _lib.check(self._lib.tacex_fem_set_indenter_mesh(self._handle, len(v), v.ctypes.data, len(t), t.ctypes.data), "tacex_fem_set_indenter_mesh")
        self.indenter_mesh = (v, t)

    def _mesh_gaps(self, x, ind):
        """torch restatement of the kernel's point-triangle distance (diagnostic; triangles in chunks to bound the memory)."""
        v, t = self.indenter_mesh
        vt = torch.from_numpy(v).to(self.device)
        r = ind[:, 5:8]
        th = r.norm(dim=-1).clamp_min(1e-300)
        K = torch.zeros((len(r), 3, 3), dtype=torch.float64, device=self.device)
        K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -r[:, 2], r[:, 1], r[:, 2], -r[:, 0], -r[:, 1], r[:, 0]
        ka = torch.where(th < 1e-12, torch.ones_like(th), torch.sin(th) / th)
        kb = torch.where(th < 1e-12, torch.zeros_like(th), (1 - torch.cos(th)) / (th * th))
        R = torch.eye(3, dtype=torch.float64, device=self.device)[None] + ka[:, None, None] * K + kb[:, None, None] * (K @ K)
        p = torch.einsum("bvj,bji->bvi", x - ind[:, None, 1:4], R)[:, :, None, :]  # R^T (x - c): (B,V,1,3)
        best = torch.full(x.shape[:2], float("inf"), dtype=torch.float64, device=self.device)
        for t0 in range(0, len(t), 16):
            tt = torch.from_numpy(t[t0:t0 + 16].astype(np.int64)).to(self.device)
            a, b, c = vt[tt[:, 0]], vt[tt[:, 1]], vt[tt[:, 2]]
            ab, ac = (b - a)[None, None], (c - a)[None, None]
            ap = p - a[None, None]
            d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
            bp = p - b[None, None]
            d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
            cp = p - c[None, None]
            d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
            vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
            den = 1.0 / (va + vb + vc)
            s, u = vb * den, vc * den
            e = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
            ue = (d4 - d3) / ((d4 - d3) + (d5 - d6))
            s, u = torch.where(e, 1 - ue, s), torch.where(e, ue, u)
            e = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
            s, u = torch.where(e, torch.zeros_like(s), s), torch.where(e, d2 / (d2 - d6), u)
            e = (d6 >= 0) & (d5 <= d6)
            s, u = torch.where(e, torch.zeros_like(s), s), torch.where(e, torch.ones_like(u), u)
            e = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
            s, u = torch.where(e, d1 / (d1 - d3), s), torch.where(e, torch.zeros_like(u), u)
            e = (d3 >= 0) & (d4 <= d3)
            s, u = torch.where(e, torch.ones_like(s), s), torch.where(e, torch.zeros_like(u), u)
            e = (d1 <= 0) & (d2 <= 0)
            s, u = torch.where(e, torch.zeros_like(s), s), torch.where(e, torch.zeros_like(u), u)
            q = a[None, None] + s[..., None] * ab + u[..., None] * ac
            best = torch.minimum(best, (p - q).norm(dim=-1).amin(-1))
        return best - ind[:, None, 4]

    def contact_gaps(self, x=None) -> torch.Tensor:
        """(num_envs, V) signed distance of every vertex to its env's indenter (+inf without one), by the solver's own distance
        function (`tacex_fem_contact_gaps`; no host round trip)."""
        if x is None:
            self.wait_for_step()  # (a step on a side stream writes self.x: the current stream reads it behind the step's event)
        x = self.x if x is None else x.to(self.device, torch.float64).contiguous()
        if getattr(self, "contact_indenters", None) is None:
            return torch.full(x.shape[:2], float("inf"), dtype=torch.float64, device=self.device)
        gaps = torch.empty(x.shape[:2], dtype=torch.float64, device=self.device)
        _lib.check(self._lib.tacex_fem_contact_gaps(self._handle, _lib.ptr(x), _lib.ptr(gaps), x.shape[0], self._stream()), "tacex_fem_contact_gaps")
        return gaps

    def contact_gaps_torch(self, x=None) -> torch.Tensor:
        """The same distances restated in torch ops (tests compare the two)."""
        x = self.x if x is None else x
        ind = getattr(self, "contact_indenters", None)
        if ind is None:
            return torch.full(x.shape[:2], float("inf"), dtype=torch.float64, device=self.device)
        c, n, kind = ind[:, None, 1:4], ind[:, None, 5:8], ind[:, None, 0]
        sph = (x - c).norm(dim=-1) - ind[:, None, 4]
        pl = ((x - c) * n).sum(-1)
        aa = (n * n).sum(-1).clamp_min(1e-300)
        t = (pl / aa).clamp(-1.0, 1.0)
        cap = (x - c - t[..., None] * n).norm(dim=-1) - ind[:, None, 4]
        inf = torch.full_like(sph, float("inf"))
        msh = self._mesh_gaps(x, ind) if getattr(self, "indenter_mesh", None) is not None and bool((ind[:, 0] == 4).any()) else inf
        return torch.where(kind == 1, sph, torch.where(kind == 2, pl, torch.where(kind == 3, cap, torch.where(kind == 4, msh, inf))))

    # -- animation targets (uipc_attachments.py:364-385) ----------------------------------------------------------
    def set_constraints(self, vertex_idx, aim_positions: torch.Tensor):
        """is_constrained[idx] = 1; aim_position[idx] = aim  (aim (B,A,3))."""
        idx = torch.as_tensor(vertex_idx, device=self.device, dtype=torch.long)
        self.is_constrained[:, idx] = 1
        self.aim_position[:, idx] = aim_positions.to(self.device, torch.float64)
        # The preconditioner depends on the constrained SET, not on the aim positions: a caller animating the targets through this
        # method every step must not trigger a host-side rebuild (device sync, dense inverse, table re-upload) per step.  The set
        # is mirrored on the host; an index tensor that lives on the device is read back once per distinct (tensor, version).
        if isinstance(vertex_idx, torch.Tensor) and vertex_idx.is_cuda:
            # (identity + version of the tensor OBJECT, which is kept alive here: the caching allocator hands the address of a freed
            #  index tensor to the next one of equal size, so a data_ptr key could take a different set for the one already mirrored)
            seen = getattr(self, "_cons_idx_seen", None)
            if seen is not None and seen[0] is vertex_idx and seen[1] == vertex_idx._version:
                return
            self._cons_idx_seen = (vertex_idx, vertex_idx._version)
            host_idx = vertex_idx.detach().reshape(-1).cpu().numpy().astype(np.int64)
        else:
            host_idx = np.asarray(vertex_idx.detach().cpu() if isinstance(vertex_idx, torch.Tensor) else vertex_idx, dtype=np.int64).reshape(-1)
        self._mark_constrained(host_idx)

    def _mark_constrained(self, host_idx):
        """Host mirror of the constrained vertex set (env 0's flags; the envs of a scene share their attachment set): the
        preconditioner is rebuilt only when a vertex joins it.  A caller that CLEARS flags by writing `is_constrained` directly
        calls `invalidate_constraints()`."""
        mirror = getattr(self, "_cons_host", None)
        if mirror is None:
            mirror = self._cons_host = np.zeros(self._obj.num_verts, dtype=bool)
        if not mirror[host_idx].all():
            mirror[host_idx] = True
            self._precond_dirty = True

    def invalidate_constraints(self):
        """After editing `is_constrained` in place (e.g. releasing vertices): the coarse operator is rebuilt from the device flags at
        the next step and the host mirror of the set starts over."""
        self._cons_host = None
        self._cons_idx_seen = None
        self._precond_dirty = True

    # -- per-env reset (uipc_object.py:280-370) ----------------------------------------------------------------------------------
    def reset(self, env_ids=None, vertex_positions: torch.Tensor | None = None):
        """Puts the gelpads of `env_ids` (None: all) back: x = `vertex_positions` ((len(env_ids), V, 3), default the rest mesh), v = 0, the
        env's step diagnostics and its friction reference cleared (`tacex_fem_reset_envs`).  The other envs are untouched; the next
        step of a reset env is the first step of a fresh scene (bit for bit with `linear_system.deterministic`).  Enqueued on the
        current stream, behind a side-stream step if there is one."""
        self.wait_for_step()
        B, V = self.num_envs, self._obj.num_verts
        ids = None
        n = B
        if env_ids is not None:
            ids = torch.as_tensor(env_ids, device=self.device).to(torch.int32).reshape(-1).contiguous()
            n = int(ids.numel())
            if n == 0:
                return
        pos = None
        if vertex_positions is not None:
            pos = vertex_positions.to(self.device, torch.float64).reshape(n, V, 3).contiguous()
        si = getattr(self, "step_info", None)
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_reset_envs(self._handle, _lib.ptr(ids), n, _lib.ptr(pos), _lib.ptr(self.x), _lib.ptr(self.v),
                                                _lib.ptr(si), _lib.ptr(self._ws), B, self._stream())
        _lib.check(rc, "tacex_fem_reset_envs")
        if self._body is not None:  # the env's free affine body goes back to where the scene placed it, at rest
            rows = slice(None) if ids is None else ids.long()
            self.q[rows] = self._q0
            self.qv[rows] = 0.0
        # (a later side-stream step is ordered behind this: FemGelpad.step makes its stream wait for the caller's before every step)

    def refresh_preconditioner(self):
        """(Re)build the coarse operator of the two-level preconditioner: Galerkin product of the REST-state matrix
        M (1 + s C) + dt^2 K_0 with the coarse space, inverted on the host (72 x 72 for the gelpad) and handed to the library.
        C = the constraint flags of env 0 (the envs of a scene share their attachment set; a preconditioner built for another
        set stays valid, only weaker).  Runs once - at the first step after `setup_sim` / `set_constraints` / the first
        `UipcIsaacAttachments.apply` - and reads the flags from the device (the only synchronisation of the FEM path)."""
        self._precond_dirty = False
        self._set_chains()
        grid = self.cfg.linear_system.coarse_grid
        if grid is None:
            _lib.check(self._lib.tacex_fem_set_coarse_space(self._handle, 0, 0, 0, 0), "tacex_fem_set_coarse_space")
            return
        from .coarse_space import build_coarse_space, coarse_grid_dims, coarse_operator_inverse

        obj = self._obj
        P, T = obj.points, obj.tets
        dims = coarse_grid_dims(P) if grid == "auto" else tuple(int(v) for v in grid)
        node, w, nc = build_coarse_space(P, dims)
        if nc > 64:
            raise ValueError(f"coarse grid {dims} has {nc} nodes (the kernel takes <= 64)")
        rest = torch.from_numpy(np.ascontiguousarray(P))[None].to(self.device)
        _, _, h = self.element_terms(rest, energy=False, gradient=False)
        He = h[0].cpu().numpy().reshape(12, 12, len(T)).transpose(2, 0, 1)
        Dm = np.stack([P[T[:, 1]] - P[T[:, 0]], P[T[:, 2]] - P[T[:, 0]], P[T[:, 3]] - P[T[:, 0]]], -1)
        det = np.linalg.det(Dm)
        T = T.copy()
        T[det < 0] = T[det < 0][:, [0, 2, 1, 3]]  # the library re-orients such tets the same way (its Hessians use that vertex order)
        vol = np.abs(det) / 6.0
        mass = np.zeros(len(P))
        np.add.at(mass, T.reshape(-1), np.repeat(obj.cfg.mass_density * vol / 4.0, 4))
        cons = self.is_constrained[0].cpu().numpy().astype(np.float64)
        aci = np.ascontiguousarray(coarse_operator_inverse(He, T, mass, cons, self._strength, self.cfg.dt, node, w, nc))
        node, w = np.ascontiguousarray(node, np.int32), np.ascontiguousarray(w, np.float64)
        _lib.check(self._lib.tacex_fem_set_coarse_space(self._handle, nc, node.ctypes.data, w.ctypes.data, aci.ctypes.data),
                   "tacex_fem_set_coarse_space")
        self.coarse_space = (node, w, aci)  # what the library was given (tests hand the same tables to the oracle)

    def _set_chains(self):
        ch = self.cfg.linear_system.vertex_chains
        if isinstance(ch, str) and ch == "auto":
            from .coarse_space import build_vertex_chains

            ch = build_vertex_chains(self._obj.points, self._obj.tets)
        ch = [list(map(int, c)) for c in (ch or []) if len(c) > 1]
        self.vertex_chains = ch  # what the library was given (tests hand the same chains to the oracle)
        if not ch:
            _lib.check(self._lib.tacex_fem_set_chains(self._handle, 0, 0, 0), "tacex_fem_set_chains")
            return
        off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(c) for c in ch])]), np.int32)
        vtx = np.ascontiguousarray(np.concatenate(ch), np.int32)
        _lib.check(self._lib.tacex_fem_set_chains(self._handle, len(ch), off.ctypes.data, vtx.ctypes.data), "tacex_fem_set_chains")

    # -- low-level entry points (thin wrappers of the C ABI) -----------------------------------------------------------
    def element_terms(self, x=None, energy=True, gradient=True, hessian=True, project_psd=False):
        x = self.x if x is None else x
        B, T = x.shape[0], self._obj.num_tets
        e = torch.empty((B, T), dtype=torch.float64, device=self.device) if energy else None
        g = torch.empty((B, 12, T), dtype=torch.float64, device=self.device) if gradient else None
        h = torch.empty((B, 144, T), dtype=torch.float64, device=self.device) if hessian else None
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_element_terms(self._handle, _lib.ptr(x), _lib.ptr(e), _lib.ptr(g), _lib.ptr(h),
                                                   1 if project_psd else 0, B, self._stream())
        _lib.check(rc, "tacex_fem_element_terms")
        return e, g, h

    def energy(self, x=None, x_tilde=None, constrained=True):
        x = self.x if x is None else x
        xt = self.x_tilde if x_tilde is None else x_tilde
        E = torch.empty((x.shape[0],), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_energy(self._handle, _lib.ptr(x), _lib.ptr(xt),
                                            _lib.ptr(self.is_constrained) if constrained else 0,
                                            _lib.ptr(self.aim_position) if constrained else 0, _lib.ptr(E),
                                            _lib.ptr(self._ws), x.shape[0], self._stream())
        _lib.check(rc, "tacex_fem_energy")
        return E

    def gradient(self, x=None, x_tilde=None, constrained=True):
        x = self.x if x is None else x
        xt = self.x_tilde if x_tilde is None else x_tilde
        g = torch.empty_like(x)
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_gradient(self._handle, _lib.ptr(x), _lib.ptr(xt),
                                              _lib.ptr(self.is_constrained) if constrained else 0,
                                              _lib.ptr(self.aim_position) if constrained else 0, _lib.ptr(g),
                                              _lib.ptr(self._ws), x.shape[0], self._stream())
        _lib.check(rc, "tacex_fem_gradient")
        return g

    def newton_step(self, constrained=True, _early_exit: bool = False):
        """One Newton iteration for every env.  `_early_exit` (used by `step`) lets envs that already converged in this time
        step return at once (device-side check, see `tacex_fem_set_newton_early_exit`)."""
        if self._precond_dirty:
            self.refresh_preconditioner()
        dx = getattr(self, "_dx", None)
        on = bool(_early_exit and dx is not None)
        if on != getattr(self, "_early_exit_on", False):
            _lib.check(self._lib.tacex_fem_set_newton_early_exit(
                self._handle, _lib.ptr(dx) if on else 0, float(self.cfg.newton.velocity_tol) * self.cfg.dt), "set_newton_early_exit")
            self._early_exit_on = on
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_newton_step(
                self._handle, _lib.ptr(self.x), _lib.ptr(self.x_tilde),
                _lib.ptr(self.is_constrained) if constrained else 0, _lib.ptr(self.aim_position) if constrained else 0,
                _lib.ptr(self.stats), _lib.ptr(self._ws), self.num_envs, int(self.cfg.linear_system.max_iter),
                float(self.cfg.linear_system.tol_rate), int(self.cfg.line_search.max_iter), self._stream())
        _lib.check(rc, "tacex_fem_newton_step")
        return self.stats

    # -- uipc_sim.py:250-252: world.advance(); world.retrieve() ---------------------------------------------------------
    def step(self, max_newton_iter: int | None = None):
        """One backward-Euler step for all envs - x_tilde = x + dt v + dt^2 g, Newton iterations, v = (x - x_n) / dt - as ONE C-ABI
        call (`tacex_fem_step`) that never touches the host: the Newton loop runs inside the kernel, every env leaves it on the
        device once the unscaled Newton direction of an iteration has max |d| <= velocity_tol * dt (uipc_sim.py:62-66).  Iteration
        counts and flags land in `self.step_info` (num_envs, 4) [newton_iterations, max |d|, flags, pcg_iterations]; reading
        `last_newton_iters` / `check_step()` is what synchronises, not the step."""
        n_max = self.cfg.newton.max_iter if max_newton_iter is None else int(max_newton_iter)
        if getattr(self, "step_info", None) is None:
            self.step_info = torch.zeros((self.num_envs, 4), dtype=torch.float64, device=self.device)
            self._g_host = (C.c_double * 3)(*[float(g) for g in self.cfg.gravity])
        if self._body is not None:  # pad + free affine body + ground: csrc/fem_ball.h (block Jacobi + the pad's coarse space + the exact ball block)
            if self._precond_dirty:
                self.refresh_preconditioner()
            with torch.cuda.device(self.device):
                rc = self._lib.tacex_fem_ball_step(
                    self._handle, _lib.ptr(self.x), _lib.ptr(self.v), _lib.ptr(self.q), _lib.ptr(self.qv), _lib.ptr(self.is_constrained),
                    _lib.ptr(self.aim_position), _lib.ptr(self.step_info), _lib.ptr(self._ball_ws), self.num_envs, self._g_host, n_max,
                    float(self.cfg.newton.velocity_tol), float(self.cfg.newton.transrate_tol), int(self.cfg.linear_system.max_iter),
                    float(self.cfg.linear_system.tol_rate), int(self.cfg.line_search.max_iter), self._stream())
            _lib.check(rc, "tacex_fem_ball_step")
            return self.x
        if self._precond_dirty:
            self.refresh_preconditioner()
        with torch.cuda.device(self.device):
            rc = self._lib.tacex_fem_step(
                self._handle, _lib.ptr(self.x), _lib.ptr(self.v), _lib.ptr(self.x_tilde), _lib.ptr(self.is_constrained),
                _lib.ptr(self.aim_position), _lib.ptr(self.stats), _lib.ptr(self.step_info), _lib.ptr(self._ws), self.num_envs,
                self._g_host, n_max, float(self.cfg.newton.velocity_tol), int(self.cfg.linear_system.max_iter),
                float(self.cfg.linear_system.tol_rate), int(self.cfg.line_search.max_iter), self._stream())
        _lib.check(rc, "tacex_fem_step")
        return self.x

    # -- a step on a side stream ---------------------------------------------------------------------------------------
    step_done = None
    """torch.cuda.Event a caller records behind a step it enqueued on a stream of its own (gelpad_scene.FemGelpad does: the FEM step and
    the optical pipeline of a sensor update are independent until the FEM-driven markers read the pad's surface, and a Newton launch
    ends with a tail of straggler envs on a mostly idle GPU - exactly where the Taxim kernels fit).  Consumers of `x` on another
    stream call `wait_for_step()` first; None: steps run on the consumers' stream."""

    def wait_for_step(self):
        """Makes the current stream wait for the last step enqueued on a side stream (no-op without one, never blocks the host)."""
        if self.step_done is not None:
            torch.cuda.current_stream(self.device).wait_event(self.step_done)

    @property
    def newton_kernel_resident(self) -> bool | None:
        """Which Newton kernel the last step launched: True = the CU-resident one (the env's state on one CU, one launch per time step), False = the
        streaming fallback (meshes whose state does not fit a CU's LDS, the deterministic switch on more than 512 vertices), None = no step yet."""
        r = int(self._lib.tacex_fem_newton_resident(self._handle))
        return None if r < 0 else bool(r)

    @property
    def last_newton_iters(self) -> int:
        """Newton iterations of the slowest env in the last step (reads the device: synchronises)."""
        si = getattr(self, "step_info", None)
        self.wait_for_step()
        return int(si[:, 0].max()) if si is not None else 0

    def check_step(self, raise_on_penetration: bool = True) -> dict:
        """Diagnostics of the last step (synchronises): flag 1 = a contact vertex was at or beyond its indenter's surface when a
        Newton iteration started - the caller moved the indenter by more than the gap between two steps (`set_contact_indenters`
        documents the contract) and that vertex gets no restoring force; flag 2 = a line search found no decrease.  Informational:
        4 = the env dropped the coarse correction for the rest of the step, 8 = it met negative curvature and finished the step with
        the PSD-safe Hessian (csrc/fem_kernels.hip, kFemFlagCoarseOff / kFemFlagPsdSafe)."""
        self.wait_for_step()  # a step enqueued on a side stream (FemGelpad(side_stream=True)): `.cpu()` only drains the CURRENT stream
        si = self.step_info.cpu().numpy()
        flags = si[:, 2].astype(np.int64)
        out = {"newton_iters": si[:, 0].astype(np.int64), "max_d": si[:, 1], "penetrating_envs": np.nonzero(flags & 1)[0],
               "line_search_failed_envs": np.nonzero(flags & 2)[0], "pcg_iters": si[:, 3].astype(np.int64),
               "coarse_dropped_envs": np.nonzero(flags & 4)[0], "psd_safe_envs": np.nonzero(flags & 8)[0],
               "pair_list_overflow_envs": np.nonzero(flags & 16)[0]}  # (affine-body scenes: a candidate list of csrc/fem_ball.h overflowed)
        if raise_on_penetration and len(out["penetrating_envs"]):
            raise RuntimeError(f"gelpad penetrated by its indenter in envs {out['penetrating_envs'][:8].tolist()}: the indenter moved by "
                               "more than the contact gap between two steps (see UipcSim.set_contact_indenters)")
        return out
