"""The gelpad scene of BASELINE configs 4 / 5 (`FemGelpad`): what a tacex_uipc task assembles from UipcSim + UipcObject +
UipcIsaacAttachments (envs/ball_rolling_uipc.py:100-140 of the reference's benchmark harness: a gelpad attached to the sensor
case, an indenter pressing into it), with the rigid bodies replaced by prescribed trajectories.  Used by bench.py and by
tests/test_fem_gpu.py; everything below runs on the device, no host round trip per step."""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .uipc_attachments import UipcIsaacAttachments, UipcIsaacAttachmentsCfg
from .uipc_object import UipcObject, UipcObjectCfg, gelpad_box_mesh
from .uipc_sim import UipcSim, UipcSimCfg


_SIDE_STREAMS: dict = {}


def _side_stream(dev, priority=0):
    """ONE side stream per device and priority for every scene of the process.  HIP deals the streams a process creates onto a handful
    of hardware queues round-robin; a side stream that lands on the queue of the caller's stream does not overlap it at all.  With a
    stream per scene, whether a scene's FEM step overlapped the optical pipeline depended on how many streams the process had created
    before it (bench.py sweep, round 5: the rolling-contact entry ran at the one-stream rate, 342 K against 388 K frames/s)."""
    d = torch.device(dev)
    key = (d.index if d.index is not None else torch.cuda.current_device(), priority)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=d, priority=priority)
    return _SIDE_STREAMS[key]


class FemGelpad:
    """C4 / C5: one ~2k-tet gelpad per env (20.75 x 25.25 x 4.5 mm block, 495 vertices / 1920 tets).  Its back face is held by
    the sensor case through UipcIsaacAttachments (aim = R(q) offset + p, soft position constraints); a spherical indenter
    presses into the front face through the IPC barrier (d_hat 1 mm, CCD-filtered Newton steps) and breathes in and out;
    stepped with UipcSim.step (backward Euler: the whole Newton loop - matrix-free PCG, CCD filter, line search - in one HIP launch)."""

    def __init__(self, B, dev, max_newton_iter: int = 8, motion: str = "breathing", side_stream: bool = False, d_hat: float | None = None,
                 cfg: UipcSimCfg | None = None, friction_lag: str | None = None, mesh: tuple[int, int, int] = (8, 10, 4)):
        """motion: "breathing" - the indenter presses in and retreats to the edge of the barrier zone every 21 steps; "rolling" - it
        stays on the pad like the ball of the reference's ball-rolling scenes: the depth varies between 0.3 and 0.8 of the env's
        maximum while the sphere slides sideways by up to +-0.5 mm (friction drags the surface along).  The half of the period in
        which the indenter RETREATS is the solver's harder regime: the pad follows it up the steeply nonlinear barrier.
        d_hat: width of the barrier zone; None = UipcSimCfg's default 1e-3 (uipc_sim.py:103-124 of the reference), 5e-4 = what the
        reference's own UIPC scenes set (ball_rolling_uipc.py:71-75, ball_rolling_tactile_rgb_uipc.py:220-224).
        cfg: a UipcSimCfg to use instead of the defaults (tolerance studies in tests/test_fem_gpu.py)."""
        assert motion in ("breathing", "rolling")
        self.motion = motion
        # side_stream: the scene driver and the FEM step run on a HIP stream of their own and `sim.step_done` marks their end, so that
        # whatever the caller enqueues next on ITS stream - the optical pipeline of the sensors' update - overlaps the Newton launch
        # (which ends with a few straggler envs on a mostly idle GPU); the FEM-driven markers wait for the event (UipcSim.wait_for_step)
        # (default stream priority: a high-priority FEM stream was measured - no gain at 320x240, 640x480 a third slower; TACEX_FEM_STREAM_PRIORITY for the A/B)
        prio = int(os.environ.get("TACEX_FEM_STREAM_PRIORITY", "0"))
        self.stream = _side_stream(dev, prio) if side_stream else None
        self.max_newton_iter = max_newton_iter
        P, T = gelpad_box_mesh(*mesh)  # (cells along x, y, z; the default is the 495-vertex / 1920-tet pad of C4 / C5)
        cfg = cfg if cfg is not None else UipcSimCfg(device=dev)
        if d_hat is not None:
            cfg.contact.d_hat = float(d_hat)
        if friction_lag is not None:
            cfg.contact.friction_lag = friction_lag
        self.d_hat = float(cfg.contact.d_hat)
        self.sim = UipcSim(cfg, num_envs=B)
        self.gelpad = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), self.sim)
        self.sim.setup_sim(constraint_strength_ratio=1000.0)  # benchmark env value (envs/ball_rolling_uipc.py:120-125)
        self.num_tets, self.num_verts = len(T), len(P)
        size = P.max(0) - P.min(0)
        body = np.array([size[0] / 2, size[1] / 2, -0.001])  # the sensor case: a plate hugging the back face
        self.att = UipcIsaacAttachments(UipcIsaacAttachmentsCfg(constraint_strength_ratio=1000.0), self.gelpad,
                                        rigid_collider=("box", (size[0] / 2 + 1e-6, size[1] / 2 + 1e-6, 0.001)), rigid_pos=body)
        self.body = torch.from_numpy(body).to(dev)
        self._body_x = float(body[0])  # (host copy: reading the device tensor per step was a device-to-host sync in the middle of the step's enqueue)
        self.quat = torch.zeros((B, 4), device=dev, dtype=torch.float64)
        self.quat[:, 0] = 1.0
        self._pos = self.body[None].repeat(B, 1).to(torch.float32).contiguous()   # per-env case position, float32 like the attachment kernel's input
        self._quat32 = self.quat.to(torch.float32).contiguous()
        top = P[:, 2].max()
        fr = np.where(P[:, 2] > top - 1e-12)[0]
        vc = fr[np.argmin(np.hypot(P[fr, 0] - size[0] / 2, P[fr, 1] - size[1] / 2))]
        self.R = 0.004
        self.z_rest = top + self.R + 0.9 * self.d_hat  # lowest point of the sphere just inside d_hat
        ind = torch.zeros((B, 8), dtype=torch.float64, device=dev)
        ind[:, 0] = 1.0
        ind[:, 1], ind[:, 2], ind[:, 3], ind[:, 4] = P[vc, 0], P[vc, 1], self.z_rest, self.R
        self.ind = ind
        self.sim.set_contact_indenters(ind)
        self.ind = self.sim.contact_indenters  # the device buffer the kernels read; moved in place every step
        self.ind0 = self.ind.clone()
        self.depth = torch.linspace(0.0004, 0.0014, B, device=dev, dtype=torch.float64)
        self._z_rest_t = torch.full((B,), self.z_rest, device=dev, dtype=torch.float64)
        self.B = B
        self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.ms_log = None    # set to [] to collect the duration of every step (hipEvents, read one step late)
        self.info_sum = None  # with ms_log: (4,) device sums over the logged steps of [Newton iterations, -, flagged envs, PCG iterations] per env mean
        self.iters_max = None  # with info_sum: device scalar, the largest Newton iteration count of any env and logged step
        self._pending = None

    def step(self, i):
        if self.stream is None:
            return self._step(i)
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(cur)  # everything enqueued so far - the previous update's marker kernel read x - is ordered before this step
        with torch.cuda.stream(self.stream):
            self._step(i)

    def _after_step(self):
        """Side stream: the event the consumers of x wait for goes right behind the step - ahead of this class's own bookkeeping kernels."""
        if self.stream is None:
            return
        if self.sim.step_done is None:
            self.sim.step_done = torch.cuda.Event()
        self.sim.step_done.record(self.stream)

    def _step(self, i):
        self.ev[0].record()
        # (the scene driver stands in for the rigid-body simulator: kept to a handful of launches - fill, attachment kernel, gap kernel +
        #  reduction, three element-wise ops - so that it does not weigh on the FEM step it is timed with)
        self._pos[:, 0].fill_(self._body_x + 0.0002 * math.sin(0.2 * i))  # the case shears the pad a little
        self.att.apply(self.sim, self._pos, self._quat32)  # compute_aim_positions -> is_constrained / aim_position (UA:364-428)
        # the indenter follows its breathing trajectory, but never moves more than half the current gap towards the pad
        # (what a CCD-filtered rigid-body step would allow); all on the device, no host round trip
        gap = self.sim.contact_gaps().amin(1)
        if self.motion == "rolling":
            c = 0.55 - 0.25 * math.cos(0.3 * i)
            dx = 0.0005 * (math.sin(0.15 * (i + 1)) - math.sin(0.15 * i))  # <= 75 um per step: far below the gap the barrier keeps
            self.ind[:, 1].add_(torch.clamp(gap * 0.25, max=abs(dx)), alpha=1.0 if dx >= 0 else -1.0)  # = clamp(dx, -gap / 4, gap / 4)
        else:
            c = 0.5 - 0.5 * math.cos(0.3 * i)
        target = torch.add(self._z_rest_t, self.depth, alpha=-c)  # z_rest - depth * c
        z = self.ind[:, 3]
        # down: limited to half the gap; up: free.  (max(target, z - gap / 2) covers both: for z <= target the second argument is below target)
        torch.maximum(target, torch.add(z, gap, alpha=-0.5), out=z)
        self.sim.step(max_newton_iter=self.max_newton_iter)
        self.ev[1].record()
        self._after_step()
        if self.ms_log is not None and self.info_sum is not None:
            self.info_sum += self.sim.step_info.mean(0)  # device-side, no synchronisation
            self.iters_max = torch.maximum(self.iters_max, self.sim.step_info[:, 0].max()) if self.iters_max is not None else self.sim.step_info[:, 0].max()
        if self.ms_log is not None:  # (reading the previous step's events: no sync with the step just enqueued)
            if self._pending is not None:
                self._pending[1].synchronize()
                self.ms_log.append(self._pending[0].elapsed_time(self._pending[1]))
            self._pending = self.ev
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def reset_indenters(self, env_ids):
        """Indenters of `env_ids` back to where the scene started them (what a task does together with `gelpad.reset(env_ids)`)."""
        self.ind[env_ids] = self.ind0[env_ids]

    def flush(self):
        """Appends the duration of the LAST enqueued step to `ms_log` (its events are read one step late; without this the log
        is shifted by one step: it would hold the step before the logged window and miss its last one).  Synchronises."""
        if self.ms_log is not None and self._pending is not None:
            self._pending[1].synchronize()
            self.ms_log.append(self._pending[0].elapsed_time(self._pending[1]))
            self._pending = None

    def fem_ms_last(self):
        if self.ms_log is not None and self._pending is None and self.ms_log:
            return self.ms_log[-1]
        ev = self._pending if (self.ms_log is not None and self._pending is not None) else self.ev
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])


def icosphere(radius: float, level: int = 2):
    """`indenter_meshes.icosphere` (12 / 42 / 162 / 642 vertices at level 0 / 1 / 2 / 3, outward oriented) turned so that a vertex - and,
    the mesh being point-symmetric, its antipode - lies on the z axis: a ball resting on the ground touches it with a vertex."""
    from .indenter_meshes import icosphere as _ico

    v, f = _ico(1.0, level)
    a = v[0]
    vx = np.cross(a, [0.0, 0.0, 1.0])
    K = np.array([[0, -vx[2], vx[1]], [vx[2], 0, -vx[0]], [-vx[1], vx[0], 0]])
    v = v @ (np.eye(3) + K + K @ K / (1.0 + a[2])).T
    return v * radius, f


class FemBallScene:
    """The reference's own UIPC scene, one per env (scripts/benchmarking/tactile_sim_performance/envs/ball_rolling_uipc.py:71-125): the gelpad
    (contact face DOWN, its back face held by the sensor case through soft position constraints, strength ratio 1000) over a FREE
    affine-body ball (m_kappa 100 MPa, density 1e3) that lies on the ground plane (ground_height 0.001), contact zone d_hat = 5e-4.  The case
    moves down and up (period 21 steps, 0.2 ... 0.8 mm of press over the envs): the pad squeezes the ball against the ground through the
    pair barriers.  Stepped by `UipcSim.step` = `tacex_fem_ball_step` (csrc/fem_ball.h).  Same driver interface as FemGelpad
    (`gelpad`, `sim`, `step(i)`, `flush()`, `ms_log`, `info_sum`, `iters_max`)."""

    def __init__(self, B, dev, max_newton_iter: int = 64, side_stream: bool = False, mesh: tuple[int, int, int] = (8, 10, 4), radius: float = 0.009,
                 level: int = 2, d_hat: float = 5e-4, ground_height: float = 0.001, ball_density: float = 1e3, cfg: UipcSimCfg | None = None,
                 shift=(0.0008, 0.0005)):
        self.stream = _side_stream(dev) if side_stream else None
        self.max_newton_iter = max_newton_iter
        P, T = gelpad_box_mesh(*mesh)
        size = P.max(0) - P.min(0)
        if cfg is None:
            cfg = UipcSimCfg(device=dev)
            # coarse grid of the two-level preconditioner: every other mesh line in x and y, one cell through the thickness - its nodes coincide
            # with mesh vertices (a nested space).  "auto" gives (2, 3, 1) cells on this pad, whose lines fall between the mesh's: measured 12.9 ->
            # 9.6 PCG iterations per Newton iteration and 6.4 -> 5.6 ms per 512-env step (scripts/r06/coarse_grid_ab.py; half the envs of this
            # scene are in light contact, where the right-hand side is smooth - inertia, the case's motion - and the coarse space does the work)
            nested = (mesh[0] // 2, mesh[1] // 2, 1)
            if mesh[0] % 2 == 0 and mesh[1] % 2 == 0 and (nested[0] + 1) * (nested[1] + 1) * 2 <= 64:
                cfg.linear_system.coarse_grid = nested
        cfg.contact.d_hat, cfg.ground_height = float(d_hat), float(ground_height)
        self.d_hat = float(d_hat)
        # the pad turned by pi about x (a rotation: the tets keep their orientation): its contact face (z = max of the box) looks down
        Pw = P * np.array([1.0, -1.0, -1.0]) + np.array([-size[0] / 2 + shift[0], size[1] / 2 + shift[1], 0.0])
        zc = ground_height + d_hat * 1.0 + radius           # the ball starts just outside the ground's barrier zone and settles into it
        Pw[:, 2] += zc + radius + 1.02 * d_hat - Pw[:, 2].min()  # the pad's face starts just outside the ball's
        # world = diag(1, -1, -1) pad + offset: the sensor camera of the upright pad (24 mm behind its back face, optical axis + z) turns with it
        self.pad_offset = Pw[0] - P[0] * np.array([1.0, -1.0, -1.0])
        self.sim = UipcSim(cfg, num_envs=B)
        self.gelpad = UipcObject(UipcObjectCfg(mesh_points=Pw, mesh_tets=T), self.sim)
        vb, tb = icosphere(radius, level)
        self.ball = UipcObject(UipcObjectCfg(mesh_points=vb, mesh_tris=tb, mass_density=ball_density, init_pos=(0.0, 0.0, zc),
                                             constitution_cfg=UipcObjectCfg.AffineBodyConstitutionCfg()), self.sim)
        self.sim.setup_sim(constraint_strength_ratio=1000.0)
        self.num_tets, self.num_verts = len(T), len(Pw)
        back = np.where(Pw[:, 2] > Pw[:, 2].max() - 1e-12)[0]
        self._back = torch.from_numpy(back).to(dev)
        self._aim0 = torch.from_numpy(Pw[back]).to(dev)[None].repeat(B, 1, 1).contiguous()
        self._aim = self._aim0.clone()
        self.depth = torch.linspace(0.0002, 0.0008, B, device=dev, dtype=torch.float64)
        self.B = B
        self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.ms_log = None
        self.info_sum = None
        self.iters_max = None
        self._pending = None

    def camera_pose(self, cam_pos_pad=(0.008, 0.012625, -0.024)):
        """(position, ROS quaternion wxyz) of the sensor camera in this scene's world frame, given its position in the upright pad's frame."""
        p = np.asarray(cam_pos_pad, np.float64) * np.array([1.0, -1.0, -1.0]) + self.pad_offset
        return tuple(float(v) for v in p), (0.0, 1.0, 0.0, 0.0)

    def step(self, i):
        if self.stream is None:
            return self._step(i)
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._step(i)

    def _step(self, i):
        self.ev[0].record()
        c = 0.5 - 0.5 * math.cos(0.3 * i)
        torch.add(self._aim0[:, :, 2], self.depth[:, None], alpha=-c, out=self._aim[:, :, 2])
        self.sim.set_constraints(self._back, self._aim)
        self.sim.step(max_newton_iter=self.max_newton_iter)
        self.ev[1].record()
        if self.stream is not None:  # (the consumers of x wait for this event: ahead of the bookkeeping kernels below)
            if self.sim.step_done is None:
                self.sim.step_done = torch.cuda.Event()
            self.sim.step_done.record(self.stream)
        if self.ms_log is not None and self.info_sum is not None:
            self.info_sum += self.sim.step_info.mean(0)
            self.iters_max = torch.maximum(self.iters_max, self.sim.step_info[:, 0].max()) if self.iters_max is not None else self.sim.step_info[:, 0].max()
        if self.ms_log is not None:
            if self._pending is not None:
                self._pending[1].synchronize()
                self.ms_log.append(self._pending[0].elapsed_time(self._pending[1]))
            self._pending = self.ev
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def flush(self):
        if self.ms_log is not None and self._pending is not None:
            self._pending[1].synchronize()
            self.ms_log.append(self._pending[0].elapsed_time(self._pending[1]))
            self._pending = None
