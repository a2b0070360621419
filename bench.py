#!/usr/bin/env python3
"""Benchmark of the tactile hot path through the drop-in boundary (GelSightSensor.update()).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Headline workload = BASELINE.json configs[2] (C3), the largest single-GPU configuration: 1024 envs x 2 GelSight Minis
(`gsmini_left` / `gsmini_right`, two independent GelSightSensor objects as in the reference's factory env,
factory_env_cfg.py:192-213) = 2048 tactile frames per step and GPU, Taxim RGB 320x240 + FOTS markers.  A "step" = one
`update(dt, force_recompute=True)` of BOTH sensors over this rank's env shard: camera depth (already resident in HBM) ->
height map + indentation depth -> Taxim RGB -> FOTS markers, then the low-resolution policy observation of both sensors
(2 x 32x32x3 uint8 + markers + indentation) is packed and collected with ONE all-gather (RCCL over xGMI when N > 1).
Weak scaling: every GPU holds 1024 envs.

At N = 1 the same process then times the other BASELINE configurations as `config.sweep[]` (not the headline value):
C2 (256 envs x 1 sensor), the 512-env shard of the 4096-env headline target, C4's per-GPU shard (512 envs RGB + markers +
the gelpad FEM step on a ~2k-tet mesh) and C5's per-GPU shard (1024 envs at 640x480 + FEM-driven markers).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries the one JSON line only)."""
    print(f"[bench {time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
FP32_PEAK_TFLOPS = 157.3
FP64_PEAK_TFLOPS = 78.6  # vector f64 (half the f32 vector rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=1024, help="env shard per GPU (weak scaling)")
    ap.add_argument("--sensors", type=int, default=2, help="GelSight sensors per env (C3: left + right finger)")
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--no-markers", action="store_true", help="Taxim RGB only (skip the FOTS marker field)")
    ap.add_argument("--gather", choices=["obs32", "none"], default="obs32",
                    help="payload of the per-step observation all-gather (obs32 = 32x32x3 RGB + markers + indentation)")
    ap.add_argument("--obs-dtype", choices=["u8", "f32"], default="u8",
                    help="dtype of the 32x32x3 policy image in the gather payload (u8 = what a CNN policy consumes)")
    ap.add_argument("--sensor-streams", action="store_true",
                    help="update the sensors of an env on one HIP stream each (+1.5 %% measured; off by default so that the per-kernel "
                         "durations of a profile of this command stay those of kernels running alone)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the C2 / 512-shard / C4 / C5 sweep (N = 1 only)")
    ap.add_argument("--sweep-steps", type=int, default=30)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=30.0, help="budget of the CPU baseline leg")
    return ap.parse_args()


def build_sensor(num_envs, H, W, markers, device, obs_res=None, obs_dtype="float32", fem_gelpad=None):
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg

    if fem_gelpad is not None:
        from tacex_amd.simulation_approaches.fem_based import ManiSkillSimulatorCfg
        # sensor camera 24 mm behind the pad's back face, optical axis along +z, marker area (x in [-8, 16.5] mm) over the pad
        marker_cfg = ManiSkillSimulatorCfg(tactile_img_res=(W, H), device=device, camera_pos_w=(0.008, 0.012625, -0.024))
    elif markers:
        marker_cfg = FOTSMarkerSimulatorCfg(tactile_img_res=(W, H), device=device)
    else:
        marker_cfg = None
    types = ["tactile_rgb", "height_map"] + (["marker_motion"] if marker_cfg is not None else [])
    cfg = GelSightSensorCfg(
        num_envs=num_envs,
        sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
        data_types=types,
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(CALIB_GELSIGHT_MINI), gelpad_height=0.0045,
                                          gelpad_to_camera_min_distance=0.024, with_shadow=False,
                                          tactile_img_res=(W, H), device=device, policy_obs_res=obs_res,
                                          policy_obs_dtype=obs_dtype),
        marker_motion_sim_cfg=marker_cfg,
        device=device,
    )
    s = GelSightSensor(cfg, gelpad_obj=fem_gelpad)
    s.initialize()
    return s


class Rig:
    """`n_sensors` GelSightSensors over one env shard + the packed observation; step() = one update of all of them."""

    def __init__(self, B, H, W, n_sensors, markers, dev, world, seed, gather="obs32", obs_dtype="u8", fem=None, sensor_streams=False):
        from tacex_amd.env_shard import ObservationGather
        from tacex_amd.utils.synthetic import synthetic_depth_maps

        self.B, self.H, self.W, self.n, self.markers, self.fem = B, H, W, n_sensors, markers, fem
        self.sensors, self.theta = [], torch.zeros(B, device=dev)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n_sensors)] if (sensor_streams and n_sensors > 1) else []
        for k in range(n_sensors):
            s = build_sensor(B, H, W, markers, dev, obs_res=(32, 32) if gather == "obs32" else None,
                             obs_dtype="uint8" if obs_dtype == "u8" else "float32",
                             fem_gelpad=fem.gelpad if fem is not None else None)
            # synthetic camera depth (metres), already resident in HBM; a different seed per shard and sensor
            hm_mm, _ = synthetic_depth_maps(B, H, W, seed=seed + 1000 * k, device=dev)
            s.set_camera_depth((hm_mm / 1000.0).contiguous())
            del hm_mm
            self.sensors.append(s)
        self.obs = None
        if gather == "obs32":
            pieces, dtypes = {}, {}
            for k in range(n_sensors):
                pieces[f"rgb32_{k}"] = (32, 32, 3)
                if obs_dtype == "u8":
                    dtypes[f"rgb32_{k}"] = torch.uint8
                pieces[f"indent_{k}"] = (1,)
                if markers or fem is not None:
                    pieces[f"markers_{k}"] = tuple(self.sensors[k]._data.output["marker_motion"].shape[1:])
            self.obs = ObservationGather(pieces, B, world, dev, dtypes=dtypes or None)

    def step(self, i=0):
        if self.fem is not None:
            self.fem.step(i)
        vals = {}
        cur = torch.cuda.current_stream()
        for k, s in enumerate(self.sensors):
            # the sensors of an env are independent objects (left / right finger): each updates on its own HIP stream, so the
            # drain of one sensor's kernels overlaps the next one's launch sequence; the packing kernel waits for all of them
            st = self.streams[k] if self.streams else cur
            if self.streams:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                if self.markers and self.fem is None:
                    s.marker_motion_simulator.set_indenter_yaw(self.theta)
                s.update(dt=0.01, force_recompute=True)
            if self.obs is not None:
                out = s._data.output
                vals[f"rgb32_{k}"] = out["tactile_rgb_obs"]  # produced inside the render pass (fused into the tail kernel)
                vals[f"indent_{k}"] = s.indentation_depth
                if self.markers or self.fem is not None:
                    vals[f"markers_{k}"] = out["marker_motion"]
        for st in self.streams:
            cur.wait_stream(st)
        if self.obs is not None:
            self.obs.pack_all(vals)
            self.obs.gather_async()  # overlaps the next step's rendering; ordered before the next pack / the final sync

    def finish(self):
        if self.obs is not None:
            self.obs.wait()  # the last step's collective is part of the timed region

    def timed(self, steps, warmup, barrier=lambda: None):
        for i in range(warmup):
            self.step(i)
        self.finish()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(warmup + i)
        self.finish()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0


class FemGelpad:
    """C4 / C5: one ~2k-tet gelpad per env (20.75 x 25.25 x 4.5 mm block, 495 vertices / 1920 tets).  Its back face is held by
    the sensor case through UipcIsaacAttachments (aim = R(q) offset + p, soft position constraints); a spherical indenter
    presses into the front face through the IPC barrier (d_hat 1 mm, CCD-filtered Newton steps) and breathes in and out;
    stepped with UipcSim.step (backward Euler: Newton + matrix-free PCG + line search in one HIP launch per Newton iteration)."""

    def __init__(self, B, dev):
        import numpy as np
        from tacex_amd.uipc import UipcIsaacAttachments, UipcIsaacAttachmentsCfg, UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg
        from tacex_amd.uipc.uipc_object import gelpad_box_mesh

        P, T = gelpad_box_mesh(8, 10, 4)
        self.sim = UipcSim(UipcSimCfg(device=dev), num_envs=B)
        self.gelpad = UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), self.sim)
        self.sim.setup_sim(constraint_strength_ratio=1000.0)  # benchmark env value (envs/ball_rolling_uipc.py:120-125)
        self.num_tets, self.num_verts = len(T), len(P)
        size = P.max(0) - P.min(0)
        body = np.array([size[0] / 2, size[1] / 2, -0.001])  # the sensor case: a plate hugging the back face
        self.att = UipcIsaacAttachments(UipcIsaacAttachmentsCfg(constraint_strength_ratio=1000.0), self.gelpad,
                                        rigid_collider=("box", (size[0] / 2 + 1e-6, size[1] / 2 + 1e-6, 0.001)), rigid_pos=body)
        self.body = torch.from_numpy(body).to(dev)
        self.quat = torch.zeros((B, 4), device=dev, dtype=torch.float64)
        self.quat[:, 0] = 1.0
        top = P[:, 2].max()
        fr = np.where(P[:, 2] > top - 1e-12)[0]
        vc = fr[np.argmin(np.hypot(P[fr, 0] - size[0] / 2, P[fr, 1] - size[1] / 2))]
        self.R = 0.004
        self.z_rest = top + self.R + 0.0009  # lowest point of the sphere just inside d_hat
        ind = torch.zeros((B, 8), dtype=torch.float64, device=dev)
        ind[:, 0] = 1.0
        ind[:, 1], ind[:, 2], ind[:, 3], ind[:, 4] = P[vc, 0], P[vc, 1], self.z_rest, self.R
        self.ind = ind
        self.sim.set_contact_indenters(ind)
        self.ind = self.sim.contact_indenters  # the device buffer the kernels read; moved in place every step
        self.depth = torch.linspace(0.0004, 0.0014, B, device=dev, dtype=torch.float64)
        self.B = B
        self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def step(self, i):
        import math
        self.ev[0].record()
        pos = self.body[None].repeat(self.B, 1)
        pos[:, 0] += 0.0002 * math.sin(0.2 * i)  # the case shears the pad a little
        self.att.apply(self.sim, pos, self.quat)  # compute_aim_positions -> is_constrained / aim_position (UA:364-428)
        # the indenter follows its breathing trajectory, but never moves more than half the current gap towards the pad
        # (what a CCD-filtered rigid-body step would allow); all on the device, no host round trip
        target = self.z_rest - self.depth * (0.5 - 0.5 * math.cos(0.3 * i))
        gap = self.sim.contact_gaps().amin(1)
        z = self.ind[:, 3]
        self.ind[:, 3] = torch.where(z > target, torch.maximum(target, z - 0.5 * gap), target)  # down: limited; up: free
        self.sim.step(max_newton_iter=8)
        self.ev[1].record()

    def fem_ms_last(self):
        self.ev[1].synchronize()
        return self.ev[0].elapsed_time(self.ev[1])


def cpu_baseline(seconds):
    """Reference CPU path (FFT-faithful torch-CPU port, oracle/taxim_torch_cpu.py) on this box's host cores.

    SURVEY 8(d) protocol: B in {1, 16, 64} at 320x240 + 8 x 640x480, median of 5 calls after 2 warm-ups, time.perf_counter.
    Thread count: the protocol's torch.set_num_threads(os.cpu_count()) is measured too (`protocol_all_cores`), but on a
    256-thread host it oversubscribes these small FFTs by two orders of magnitude (one 320x240 frame took 21 s), so `value` is
    the BEST thread count of a sweep {8, 16, 32, 64, physical cores} - the baseline is never handicapped by the thread setting -
    and `cores` is the thread count that produced it."""
    from oracle.taxim_torch_cpu import TaximTorchCpuPort
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    ncpu = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        phys = None
    t_start = time.perf_counter()
    left = lambda: seconds - (time.perf_counter() - t_start)
    ports = {}

    def calls(H, W, B, reps, warm):
        if (H, W) not in ports:
            ports[(H, W)] = TaximTorchCpuPort(CALIB_GELSIGHT_MINI, (H, W))
        hm, ind = synthetic_depth_maps(B, H, W, seed=1, device="cpu")
        ts = []
        for r in range(warm + reps):
            t0 = time.perf_counter()
            ports[(H, W)].render_direct(hm, ind)
            dt = time.perf_counter() - t0
            if r >= warm:
                ts.append(dt)
            if dt * 2 > left() and ts:
                break
        return ts

    # 1) thread sweep at B = 16
    sweep_t = []
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, phys or 64)}):
        torch.set_num_threads(th)
        ts = calls(240, 320, 16, 3, 1)
        sweep_t.append({"threads": th, "frames_per_s": round(16 / statistics.median(ts), 2)})
        # past the sweet spot the small FFTs oversubscribe and the rate collapses (64 threads: 10 frames/s on this host): stop there
        if left() < seconds * 0.6 or sweep_t[-1]["frames_per_s"] < 0.7 * max(e["frames_per_s"] for e in sweep_t):
            break
    best_th = max(sweep_t, key=lambda e: e["frames_per_s"])["threads"]
    log(f"cpu baseline thread sweep: {sweep_t} -> {best_th}")
    # 2) the protocol's batch sizes at the best thread count
    torch.set_num_threads(best_th)
    runs, best = [], 0.0
    for (H, W, B) in ((240, 320, 1), (240, 320, 16), (240, 320, 64), (480, 640, 8)):
        if left() < 2.0:
            break
        ts = calls(H, W, B, 5, 2)
        med = statistics.median(ts)
        runs.append({"frames": B, "resolution": [W, H], "threads": best_th, "median_s": round(med, 4),
                     "frames_per_s": round(B / med, 2), "timed_calls": len(ts)})
        if (H, W) == (240, 320):
            best = max(best, B / med)
        log(f"cpu baseline {W}x{H} B={B} threads={best_th}: {B / med:.1f} frames/s ({len(ts)} calls)")
    # 3) the protocol's thread setting (all logical cores), one bounded sample
    proto = None
    if ncpu != best_th and left() > 1.0:
        torch.set_num_threads(ncpu)
        t0 = time.perf_counter()
        ts = calls(240, 320, 16, 1, 0)
        proto = {"threads": ncpu, "frames": 16, "frames_per_s": round(16 / ts[0], 3), "timed_calls": 1,
                 "note": "torch.set_num_threads(os.cpu_count()), first call (thread-pool start-up included)"}
        log(f"cpu baseline protocol all-cores: {proto}")
        torch.set_num_threads(best_th)
    return {"value": round(best, 2), "unit": "frames/s", "cores": best_th, "logical_cores": ncpu, "physical_cores": phys, "kind": "port",
            "sample": "Taxim RGB no-shadow (reflect-pad + torch.fft correlation x7, gather + polynomial; oracle/taxim_torch_cpu.py) on the "
                      "same synthetic depth maps (seed 1); value = best 320x240 batch size of {1, 16, 64} at the best intra-op thread "
                      "count of the sweep, median of 5 calls after 2 warm-ups",
            "runs": runs, "thread_sweep_B16": sweep_t, "protocol_all_cores": proto, "wall_s": round(time.perf_counter() - t_start, 1)}


def roofline_leg(rig, markers):
    """Per-stage hipEvent timing on the launch stream (library-side events, tacex_taxim_set_profiling), outside the timed
    region.  Durations are per LAUNCH; a launch covers `chunk` frames (large shards are walked in Infinity-Cache-sized chunks)."""
    s = rig.sensors[0]
    H, W, B = rig.H, rig.W, rig.B
    taxim = s.optical_simulator._taxim
    chunk = taxim.chunk_frames((H, W), B)
    taxim.set_profiling((H, W), True)
    for _ in range(10):
        s.update(dt=0.01, force_recompute=True)
    torch.cuda.synchronize()
    prof = taxim.read_profile((H, W))
    taxim.set_profiling((H, W), False)
    N = H * W
    stages = {}
    for name, (ms, cnt) in prof.items():
        if cnt == 0:
            continue
        frames = chunk
        if name == "frame_min":
            bpf, frames = 4 * N, B  # the shard-wide reduction pass runs once over all B frames when the shard is chunked
            if chunk == B:
                continue  # one-pass shards get the minimum from the fused depth -> height-map kernel (not a Taxim stage)
        elif name.startswith("blur_l0"):
            bpf = 8 * N          # read height map, write level 0
        elif name.startswith("blur_"):
            bpf = 12 * N         # read previous level + height map (masked restore), write level
        elif name.startswith("tail"):
            # read level + height map, write RGB; FOTS gets the marker pixels + per-wave statistics from the same kernel
            # (a few KB per frame), so the full deformed-gel / mask frames (5 B/px) are only stored on the fallback path
            full_frames = markers and getattr(s.optical_simulator, "_fots_compact_version", -1) < 0
            bpf = (20 + (5 if full_frames else 0)) * N
        else:
            bpf = 16 * N         # shade: read deformed gel, write RGB
        avg = ms / cnt
        if name.startswith("blur_") and cnt > 10 * max(1, B // chunk):
            frames = B * 10 // cnt  # band levels run over Infinity-Cache-sized sub-ranges of a pass: frames per launch from the launch count
        stages[name] = {"avg_ms": round(avg, 5), "frames_per_launch": frames, "algo_bytes_per_launch": bpf * frames,
                        "GBps": round(bpf * frames / (avg * 1e-3) / 1e9, 1), "launches_per_update": round(cnt / 10, 2)}
    dom = max((k for k in stages if k != "frame_min"), key=lambda k: stages[k]["avg_ms"] * stages[k]["launches_per_update"])
    taxim_ms = sum(st["avg_ms"] * st["launches_per_update"] for st in stages.values())  # per update of B frames
    ach = stages[dom]["GBps"]
    pipeline_gbs = 16 * N * B / (taxim_ms * 1e-3) / 1e9
    tb = taxim.context((H, W)).tables
    flops_per_frame = 2 * sum(kw + kh for kw, kh in zip(tb.ksize_w, tb.ksize_h)) * N + 100 * N
    # SURVEY 8(d) figure for the dominant kernel: 16 B/px (read height map + write RGB) x frames per launch
    survey_gbs = 16 * N * stages[dom]["frames_per_launch"] / (stages[dom]["avg_ms"] * 1e-3) / 1e9
    roof = {
        "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
        "frames_per_launch": stages[dom]["frames_per_launch"],
        "achieved_survey_16B_per_px": round(survey_gbs, 1), "frac_survey_16B_per_px": round(survey_gbs / HBM_PEAK_GBS, 4),
        "pipeline_achieved": round(pipeline_gbs, 1), "pipeline_frac": round(pipeline_gbs / HBM_PEAK_GBS, 4),
        "valu_frac": round(flops_per_frame * B / (taxim_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
        "stages": stages,
        "note": "achieved = algorithmic bytes of the dominant kernel per launch (its own reads + writes, DESIGN.md section 4) / its "
                "hipEvent-measured duration; *_survey_16B_per_px = SURVEY 8(d)'s 16 B/px x frames per launch / the same duration; "
                "pipeline_* = 16 B/px compulsory bytes of the whole Taxim path / sum of its kernels; valu_frac = algorithmic fp32 "
                "flops / 157.3 TFLOP/s (the separable blur is VALU-heavy)",
    }
    # HBM bytes of the dominant kernel per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate `rocprofv3 --pmc` runs,
    # committed under profiles/ - counters cannot be read live from inside the process, so this is a build-time constant)
    for cand in ("pmc_traffic_r02.json", "pmc_traffic.json"):
        pmc = REPO / "profiles" / cand
        if pmc.exists() and (H, W) == (240, 320):
            try:
                j = json.loads(pmc.read_text())
                per_frame = j["per_frame_bytes"].get(dom)
                if per_frame is not None:
                    roof["traffic"] = int(per_frame * stages[dom]["frames_per_launch"])
                    roof["traffic_source"] = f"profiles/{cand} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per-frame bytes x frames per launch)"
                    roof["traffic_measured_at"] = j.get("measured_at_commit", "see _provenance")
                    break
            except Exception:
                pass
    return roof


def sweep(args, dev):
    """The other BASELINE configurations, timed the same way (rank 0, N = 1, after the headline)."""
    out = []

    def run(label, B, H, W, n_sensors, markers, fem=None, steps=None):
        steps = steps or args.sweep_steps
        log(f"sweep: {label}")
        try:
            if callable(fem):
                fem = fem()
            rig = Rig(B, H, W, n_sensors, markers, dev, 1, seed=7, gather=args.gather, obs_dtype=args.obs_dtype, fem=fem)
            el = rig.timed(steps, 3)
            frames = B * n_sensors * steps
            e = {"workload": label, "frames_per_step": B * n_sensors, "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
                 "frames_per_s": round(frames / el, 1)}
            if fem is not None:
                # split: the FEM step alone (events around attachments + UipcSim.step of the last step)
                e["fem_ms_last_step"] = round(fem.fem_ms_last(), 3)
                e["fem_newton_iters_last_step"] = int(fem.sim.last_newton_iters)
                e["fem"] = fem_roofline(fem)
            out.append(e)
            del rig
            torch.cuda.empty_cache()
        except Exception as ex:  # a sweep entry must not take the headline line down with it
            out.append({"workload": label, "error": f"{type(ex).__name__}: {ex}"[:300]})

    run("C2: 256 envs x 1 sensor, Taxim RGB 320x240 (BASELINE configs[1])", 256, 240, 320, 1, False)
    run("C2 + FOTS markers: 256 envs x 1 sensor, RGB 320x240 + markers", 256, 240, 320, 1, True)
    run("512-env shard of the 4096-env / 8-GPU target: 512 envs x 1 sensor, RGB 320x240 + FOTS markers", 512, 240, 320, 1, True)
    run("C4 per-GPU shard: 512 envs, RGB 320x240 + FEM-driven markers + gelpad FEM step (1920 tets / env) (BASELINE configs[3] / 8)",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev), steps=max(12, args.sweep_steps // 3))
    run("C5 per-GPU shard: 1024 envs, RGB 640x480 + FEM-driven markers (gelpad FEM step included) (BASELINE configs[4] / 8)",
        1024, 480, 640, 1, False, fem=lambda: FemGelpad(1024, dev), steps=max(12, args.sweep_steps // 6))
    run("C5 optical part only: 1024 envs, RGB 640x480", 1024, 480, 640, 1, False, steps=max(5, args.sweep_steps // 3))
    return out


def fem_roofline(fem):
    """Roofline of the FEM inner step (1 456 B per tet and Newton iteration assembled, 304 B matrix-free)."""
    sim = fem.sim
    B, T = fem.B, fem.num_tets

    def timeit(fn, n=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / n

    ms_el = timeit(lambda: sim.element_terms())
    x0 = sim.x.clone()

    def newton():
        sim.x.copy_(x0)
        sim.newton_step()

    ms_nw = timeit(newton, 3)
    st = sim.stats.cpu().numpy()
    pcg = float(st[:, 3].mean())
    sim.x.copy_(x0)
    return {
        "element_terms": {"bound": "hbm", "ms": round(ms_el, 4), "algo_bytes": 1464 * B * T, "achieved": round(1464 * B * T / (ms_el * 1e-3) / 1e9, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(1464 * B * T / (ms_el * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "assembled energy + gradient + 12x12 Hessian per tet: 208 B read + 8 + 96 + 1152 B written (fp64)"},
        "newton_iteration": {"ms": round(ms_nw, 3), "pcg_iterations_mean": round(pcg, 1),
                             "matrix_free_GBps_304B_per_tet": round(304 * B * T / (ms_nw * 1e-3) / 1e9, 2),
                             "note": "one launch = gradient + block preconditioner + matrix-free PCG + line search with the env's state "
                                     "resident on the CU (LDS / registers): no HBM traffic inside the PCG loop, so the 304 B/tet matrix-free "
                                     "figure is a per-Newton-iteration lower bound, not the binding roof (f64 latency-bound, DESIGN.md section 4)"},
    }


def main():
    args = parse()
    from tacex_amd import _lib
    from tacex_amd.env_shard import init_from_env

    # Native libraries write banners to the C stdout (RCCL prints its version block when the first communicator is made, and
    # libc only flushes it at exit - after our line).  The driver parses stdout for ONE JSON line: everything before it goes to
    # stderr instead (fd 1 -> fd 2 until the line is printed).
    import ctypes
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    H, W = args.height, args.width
    markers = not args.no_markers
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...` "
                             f"(WORLD_SIZE={world})")
    shard = init_from_env(args.envs_per_gpu * args.gpus,
                          backend="nccl" if (args.gpus > 1 or os.environ.get("TACEX_FORCE_DIST") == "1") else None)
    dev = f"cuda:{shard.local_rank}"
    torch.cuda.set_device(shard.local_rank)
    B = shard.num_local
    use_dist = dist.is_available() and dist.is_initialized()

    def barrier():
        if use_dist:
            dist.barrier()

    log(f"headline: {B} envs x {args.sensors} sensors, {W}x{H}, rank {shard.rank}/{shard.world_size}")
    rig = Rig(B, H, W, args.sensors, markers, dev, shard.world_size, seed=1 + shard.rank, gather=args.gather,
              obs_dtype=args.obs_dtype, sensor_streams=args.sensor_streams)
    elapsed = rig.timed(args.steps, args.warmup, barrier)
    log(f"headline timed: {elapsed / args.steps * 1e3:.3f} ms/step")
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames_per_step = args.envs_per_gpu * args.sensors * args.gpus
    value = frames_per_step * args.steps / elapsed

    roofline = None
    if not args.no_roofline and shard.rank == 0:
        roofline = roofline_leg(rig, markers)
    obs_bytes = None if rig.obs is None else rig.obs.payload_bytes()
    sensor_streams_on = bool(rig.streams)
    del rig
    torch.cuda.empty_cache()

    sw = None
    if args.gpus == 1 and not args.no_sweep and shard.rank == 0 and not use_dist:
        sw = sweep(args, dev)

    cpu = None
    if not args.no_cpu_baseline and shard.rank == 0 and args.gpus == 1:
        log("cpu baseline leg")
        cpu = cpu_baseline(args.cpu_baseline_seconds)
    log("done")

    if shard.rank == 0:
        line = {
            "metric": "tactile_frames_per_sec", "value": round(value, 1), "unit": "frames/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.envs_per_gpu} envs x {args.sensors} GelSight Mini per GPU = {args.envs_per_gpu * args.sensors} frames/step/GPU, "
                            f"Taxim RGB {W}x{H}" + (" + FOTS markers (99)" if markers else "")
                            + f" via {args.sensors} GelSightSensor.update() calls per step"
                            + ("; BASELINE configs[2] (C3)" if (args.envs_per_gpu, args.sensors, W, H, markers) == (1024, 2, 320, 240, True) else ""),
                "envs_per_gpu": args.envs_per_gpu, "sensors_per_env": args.sensors, "frames_per_step": frames_per_step,
                "resolution": [W, H], "markers": markers,
                "two_sensor_batching": "two independent GelSightSensor objects (gsmini_left / gsmini_right as factory_env_cfg.py:192-213), "
                                       f"each one launch sequence over its {args.envs_per_gpu} envs"
                                       + (", one HIP stream per sensor (joined before the observation is packed)" if sensor_streams_on else " on the same stream"),
                "observation_gather": None if obs_bytes is None else {
                    "payload": f"per sensor: 32x32x3 {args.obs_dtype} RGB (antialiased, produced in the render pass) + f32 indentation"
                               + (" + f32 markers (2,99,2)" if markers else ""),
                    "bytes_per_rank": obs_bytes,
                    "collective": "all_gather_into_tensor x1 per step" if (args.gpus > 1 or use_dist) else "none (N=1: the packed buffer is the observation)"},
                "background_frame": "synthetic f0 (real dataPack.npz absent from the reference checkout)",
                "arch": _lib.require_gpu(shard.local_rank),
            },
        }
        if sw is not None:
            line["config"]["sweep"] = sw
        if roofline is not None:
            line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = cpu
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    try:
        ctypes.CDLL(None).fflush(None)  # native buffers (the RCCL banner) land on stderr
    except OSError:
        pass
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if shard.rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
