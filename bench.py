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

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries the one JSON line only)."""
    print(f"[bench {time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
FP32_PEAK_TFLOPS = 157.3
F64_PEAK_TFLOPS = 78.6   # MI355X vector fp64 (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
LDS_PEAK_TBS = 78.6      # 128 B/clk/CU x 256 CUs x 2.4 GHz (guides/MI355X_MICROARCH.md, LDS section)
FP64_PEAK_TFLOPS = 78.6  # vector f64 (half the f32 vector rate)


def parse(argv=None):
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
                    help="update the sensors of an env on one HIP stream each (+3.3 %% measured in round 3: 660 K vs 638 K frames/s; off by "
                         "default so that the per-kernel durations of a profile of this command stay those of kernels running alone - under "
                         "overlap rocprofv3 reports 1084 us for a tail launch that takes 762 us alone)")
    ap.add_argument("--no-group", action="store_true",
                    help="update the sensors of an env one by one (two launch sequences per step) instead of as one GelSightSensorGroup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-node-leg", action="store_true", help="skip the 4096-env whole-node leg (value_node4096*)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the C2 / 512-shard / C4 / C5 sweep (N = 1 only)")
    ap.add_argument("--sweep-steps", type=int, default=30)
    ap.add_argument("--sweep-keys", default=None, help="comma-separated sweep entries to run (default: all), in the order given by the sweep")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=30.0, help="budget of the CPU baseline leg")
    ap.add_argument("--details-out", default=None,
                    help="side file for everything that is not the headline (sweep entries with their descriptions, per-stage timings, CPU "
                         "thread sweep); default gpurun_out/bench_details_n<N>.json.  stdout carries ONE compact JSON line only")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="no GPU: every rank runs the launch / rendezvous / max-over-ranks timing / emitter path over gloo with a stand-in "
                         "step (tests/test_env_shard_gloo.py); the printed line is marked data = 'dry-run' and is not a measurement")
    return ap.parse_args(argv)


def build_sensor(num_envs, H, W, markers, device, obs_res=None, obs_dtype="float32", fem_gelpad=None, cam_res=None, clip=(0.024, 0.029),
                 grid=(11, 9), initialize=True, cam_pose=None):
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg

    if fem_gelpad is not None:
        from tacex_amd.simulation_approaches.fem_based import ManiSkillSimulatorCfg
        # sensor camera 24 mm behind the pad's back face, optical axis along +z, marker area (x in [-8, 16.5] mm) over the pad
        marker_cfg = ManiSkillSimulatorCfg(tactile_img_res=(W, H), device=device, camera_pos_w=(0.008, 0.012625, -0.024))
        if cam_pose is not None:  # a scene that places the pad elsewhere (FemBallScene: contact face down over the ball)
            marker_cfg.camera_pos_w, marker_cfg.camera_quat_w_ros = cam_pose
    elif markers:
        marker_cfg = FOTSMarkerSimulatorCfg(tactile_img_res=(W, H), device=device,
                                            marker_params=FOTSMarkerSimulatorCfg.MarkerParams(num_markers_col=grid[0], num_markers_row=grid[1], x0=15, y0=26))
    else:
        marker_cfg = None
    types = ["tactile_rgb", "height_map"] + (["marker_motion"] if marker_cfg is not None else [])
    cfg = GelSightSensorCfg(
        num_envs=num_envs,
        sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=cam_res or (W, H), clipping_range=clip),
        data_types=types,
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(CALIB_GELSIGHT_MINI), gelpad_height=0.0045,
                                          gelpad_to_camera_min_distance=0.024, with_shadow=False,
                                          tactile_img_res=(W, H), device=device, policy_obs_res=obs_res,
                                          policy_obs_dtype=obs_dtype),
        marker_motion_sim_cfg=marker_cfg,
        device=device,
    )
    s = GelSightSensor(cfg, gelpad_obj=fem_gelpad)
    if initialize:
        s.initialize()
    return s


class Rig:
    """`n_sensors` GelSightSensors over one env shard + the packed observation; step() = one update of all of them."""

    def __init__(self, B, H, W, n_sensors, markers, dev, world, seed, gather="obs32", obs_dtype="u8", fem=None, sensor_streams=False,
                 data="contacts", cam_res=None, clip=(0.024, 0.029), grid=(11, 9), group=True, collective=None):
        from tacex_amd import GelSightSensorGroup
        from tacex_amd.env_shard import ObservationGather
        from tacex_amd.utils.synthetic import dense_contact_depth_maps, synthetic_depth_maps

        self.B, self.H, self.W, self.n, self.markers, self.fem = B, H, W, n_sensors, markers, fem
        self.sensors, self.theta = [], torch.zeros(B, device=dev)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n_sensors)] if (sensor_streams and n_sensors > 1) else []
        # the sensors of an env (left / right finger) share one configuration: grouped, their frames are ONE 2B-frame launch sequence
        # (GelSightSensorGroup).  group=False / one stream per sensor: two independent updates per step
        self.grouped = bool(group and n_sensors > 1 and fem is None and not self.streams)
        built = [build_sensor(B, H, W, markers, dev, obs_res=(32, 32) if gather == "obs32" else None,
                              obs_dtype="uint8" if obs_dtype == "u8" else "float32",
                              fem_gelpad=fem.gelpad if fem is not None else None, cam_res=cam_res, clip=clip, grid=grid,
                              initialize=not self.grouped, cam_pose=fem.camera_pose() if hasattr(fem, "camera_pose") else None)
                 for _ in range(n_sensors)]
        self.group = GelSightSensorGroup(built) if self.grouped else None
        for k in range(n_sensors):
            s = built[k]
            # synthetic camera depth (metres), already resident in HBM; a different seed per shard and sensor
            Wc, Hc = cam_res or (W, H)
            gen = dense_contact_depth_maps if data == "dense" else synthetic_depth_maps
            hm_mm, _ = gen(B, Hc, Wc, seed=seed + 1000 * k, device=dev)
            if clip[1] > 0.029:  # a scene whose far plane lies behind the gel: the camera sees nothing there (inf, GS:581-588)
                hm_mm = torch.where(hm_mm >= 29.0, torch.full_like(hm_mm, float("inf")), hm_mm)
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
            self.obs = ObservationGather(pieces, B, world, dev, dtypes=dtypes or None, collective=collective)

    def step(self, i=0):
        if self.fem is not None:
            self.fem.step(i)
        vals = {}
        cur = torch.cuda.current_stream()
        if self.grouped and self.markers:  # inputs of every member first: the first member's update evaluates the whole group
            for s in self.sensors:
                # (the yaw, like the depth image, is resident where the sensor reads it: the member's rows of the group's yaw vector -
                #  a producer writes there; handing over a separate tensor costs one 4 KB device copy per member and step)
                s.marker_motion_simulator.set_indenter_yaw(s.marker_motion_simulator.indenter_yaw_buffer())
        for k, s in enumerate(self.sensors):
            # the sensors of an env are independent objects (left / right finger): each updates on its own HIP stream, so the
            # drain of one sensor's kernels overlaps the next one's launch sequence; the packing kernel waits for all of them
            st = self.streams[k] if self.streams else cur
            if self.streams:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                if self.markers and self.fem is None and not self.grouped:
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
        if self.fem is not None:
            self.fem.flush()  # the last step's hipEvent pair joins ms_log: entry k of the log IS step k

    def timed(self, steps, warmup, barrier=lambda: None, after_warmup=lambda: None, windows=1):
        """Wall time of `steps` steps (after `warmup` untimed ones).  windows > 1: the steps are timed as that many consecutive windows
        with a synchronisation between them; the per-window times land in `self.window_s` (a runtime stall inside one window shows)."""
        import gc

        for i in range(warmup):
            self.step(i)
        self.finish()
        torch.cuda.synchronize()
        after_warmup()
        # (no cyclic-garbage collection inside the timed region, like timeit: a full collection of this process - torch's module graph -
        #  stalls the enqueuing thread for 60-70 ms, measured as a "76 ms step" once per few hundred steps of scripts/fem_stress.py)
        gc.collect()
        gc.disable()
        try:
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.window_s, per = [], steps // windows
            for w in range(windows):
                tw = time.perf_counter()
                for i in range(w * per, steps if w == windows - 1 else (w + 1) * per):
                    self.step(warmup + i)
                self.finish()
                torch.cuda.synchronize()
                self.window_s.append(time.perf_counter() - tw)
            barrier()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        finally:
            gc.enable()


from tacex_amd.uipc.gelpad_scene import FemBallScene, FemGelpad  # noqa: E402  (C4 / C5: the gelpad scene lives in the package, tests step it too)


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
    # 3) the protocol's thread setting (all logical cores): ONE 320x240 frame in a child process that is ended after 5 s (on a 256-thread
    #    host the oversubscribed small FFTs took 21 s for that frame - 30 of the 58 s of round 4's driver run)
    proto = None
    if ncpu != best_th and left() > 1.0:
        proto = protocol_all_cores(ncpu, cap_s=5.0)
        log(f"cpu baseline protocol all-cores: {proto}")
    return {"value": round(best, 2), "unit": "frames/s", "cores": best_th, "logical_cores": ncpu, "physical_cores": phys, "kind": "port",
            "sample": "Taxim RGB no-shadow (reflect-pad + torch.fft correlation x7, gather + polynomial; oracle/taxim_torch_cpu.py) on the "
                      "same synthetic depth maps (seed 1); value = best 320x240 batch size of {1, 16, 64} at the best intra-op thread "
                      "count of the sweep, median of 5 calls after 2 warm-ups",
            "runs": runs, "thread_sweep_B16": sweep_t, "protocol_all_cores": proto, "wall_s": round(time.perf_counter() - t_start, 1)}


def protocol_all_cores(ncpu, cap_s=5.0):
    """SURVEY 8(d)'s literal thread setting, `torch.set_num_threads(os.cpu_count())`, on one 320x240 frame - in a child process
    with a hard cap of `cap_s` seconds of compute (a call cannot be interrupted from inside)."""
    import subprocess

    code = ("import sys, time; sys.path.insert(0, %r); import torch; torch.set_num_threads(%d)\n"
            "from oracle.taxim_torch_cpu import TaximTorchCpuPort\n"
            "from tacex_amd.calibration import CALIB_GELSIGHT_MINI\n"
            "from tacex_amd.utils.synthetic import synthetic_depth_maps\n"
            "p = TaximTorchCpuPort(CALIB_GELSIGHT_MINI, (240, 320)); hm, ind = synthetic_depth_maps(1, 240, 320, seed=1, device='cpu')\n"
            "print('READY', flush=True); t0 = time.perf_counter(); p.render_direct(hm, ind); print('T', time.perf_counter() - t0, flush=True)\n"
            % (str(REPO), ncpu))
    out = {"threads": ncpu, "frames": 1, "cap_s": cap_s, "note": "torch.set_num_threads(os.cpu_count()), first call (thread-pool start-up included)"}
    try:
        pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        ready = pr.stdout.readline()  # tables loaded, the timed call starts now
        if not ready.startswith("READY"):
            raise RuntimeError("child did not start")
        try:
            rest, _ = pr.communicate(timeout=cap_s)
            t1 = float(rest.split()[1])
            out.update({"frames_per_s": round(1 / t1, 3), "one_frame_call_s": round(t1, 3)})
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.communicate()
            out.update({"frames_per_s": None, "did_not_finish_within_cap": True, "frames_per_s_upper_bound": round(1 / cap_s, 3)})
    except Exception as ex:
        out["error"] = f"{type(ex).__name__}: {ex}"[:200]
    return out


def roofline_leg(rig, markers):
    """Per-stage hipEvent timing on the launch stream (library-side events, tacex_taxim_set_profiling), outside the timed
    region.  Durations are per LAUNCH; a launch covers `chunk` frames (large shards are walked in Infinity-Cache-sized chunks)."""
    s = rig.sensors[0]  # (a grouped member: its update evaluates the group's core sensor of n x B frames)
    H, W, B = rig.H, rig.W, (rig.group.core._num_envs if rig.grouped else rig.B)
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
    # `achieved` / `frac` follow SURVEY 8(d) to the letter: ALGORITHMIC bytes = 16 B/px (read the height map, write RGB) x the pixels
    # one launch of the dominant kernel processes, divided by that kernel's hipEvent-measured average launch duration.  The kernel's
    # OWN reads + writes (20 B/px for the fused tail: it also reads the previous pyramid level) are kept beside it as *_own_bytes.
    roof = {
        "bound": "hbm", "kernel": dom, "achieved": round(survey_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(survey_gbs / HBM_PEAK_GBS, 4), "traffic": None,
        "frames_per_launch": stages[dom]["frames_per_launch"],
        "kernel_avg_ms": stages[dom]["avg_ms"],
        "algorithmic_bytes_per_launch": 16 * N * stages[dom]["frames_per_launch"],
        "achieved_own_bytes": ach, "frac_own_bytes": round(ach / HBM_PEAK_GBS, 4),
        "pipeline_achieved": round(pipeline_gbs, 1), "pipeline_frac": round(pipeline_gbs / HBM_PEAK_GBS, 4),
        "valu_frac": round(flops_per_frame * B / (taxim_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
        "stages": stages,
        "note": "achieved = SURVEY 8(d)'s 16 B/px x the pixels of one launch of the dominant kernel / its hipEvent-measured average "
                "duration (frac = achieved / 8 TB/s; reproduces from profiles/r06_c3_kernel_stats.csv: 16 x 76800 x frames per launch B / the tail's "
                "average duration); *_own_bytes = the same with the kernel's own reads + writes (DESIGN.md section 4); pipeline_* = "
                "16 B/px of the whole Taxim path / sum of its kernels; valu_frac = algorithmic fp32 flops / 157.3 TFLOP/s (the "
                "separable blur is VALU-heavy)",
    }
    # HBM bytes of the dominant kernel per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate `rocprofv3 --pmc` runs,
    # committed under profiles/ - counters cannot be read live from inside the process, so this is a build-time constant)
    for cand in ("pmc_traffic_r06.json", "pmc_traffic_r05.json", "pmc_traffic_r04.json", "pmc_traffic_r03.json", "pmc_traffic_r02.json", "pmc_traffic.json"):
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

    only = set(args.sweep_keys.split(",")) if args.sweep_keys else None

    def run(key, label, B, H, W, n_sensors, markers, fem=None, steps=None, gather=None, count_in_contact=False, cap_may_bind=False, **rig_kw):
        if only is not None and key not in only:
            return
        steps = steps or args.sweep_steps
        log(f"sweep: {key}: {label}")
        try:
            # (every entry starts from a collected heap and an empty allocator cache: the previous entry's rig is garbage with
            #  reference cycles, and its buffers are gigabytes)
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            if callable(fem):
                fem = fem()
            rig = Rig(B, H, W, n_sensors, markers, dev, 1, seed=7, gather=gather or args.gather, obs_dtype=args.obs_dtype, fem=fem, **rig_kw)
            if fem is not None:
                fem.ms_log = []
            # FEM scenes: the indenter breathes with a period of 2 pi / 0.3 = 21 steps whose phases cost 1 ... 18 ms each, and the first
            # long Newton launches of a process carry a one-time ~50 ms of runtime work (scratch set-up, seen at sync; scripts/c4_probe.py):
            # warm up over one whole period, time exactly THREE periods as three windows (a single period is ~27 ms of wall time: one
            # runtime stall of 20-50 ms - one in ~13 processes carried one, profiles/r05_experiments.md section 13 - halved the entry; the
            # rate is still the mean over all 63 steps, the per-period times are in the entry)
            if fem is not None:
                steps = 63
            base = []
            if fem is not None:  # solver statistics over the timed period: device-side sums, the warm-up's share is subtracted
                fem.info_sum = torch.zeros(4, dtype=torch.float64, device=dev)
            el = rig.timed(steps, 24 if fem is not None else 8, after_warmup=lambda: base.append(fem.info_sum.clone()) if fem is not None else None,
                           windows=3 if fem is not None else 1)
            frames = B * n_sensors * steps
            e = {"key": key, "workload": label, "frames_per_step": B * n_sensors, "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
                 "frames_per_s": round(frames / el, 1)}
            if count_in_contact:
                # the reference's counting rule (run_ball_rolling_experiment.py:238-244): only frames with indentation_depth > 0
                # count as rendered tactile frames (its harness skips the rest); the inputs are static, so one read after the run
                inc = sum(int((s_.indentation_depth > 0).sum()) for s_ in rig.sensors)
                e["frames_in_contact_per_step"] = inc
                e["frames_per_s_in_contact_only"] = round(inc * steps / el, 1)
            if fem is not None:
                e["period_ms"] = [round(w * 1e3, 2) for w in rig.window_s]  # wall time of each 21-step period of the indenter's motion
                # split: the FEM part alone (hipEvents around attachments + UipcSim.step), MEAN over the timed steps - the Newton
                # / PCG iteration counts vary from step to step with the indenter's breathing
                ms = fem.ms_log[24:] or fem.ms_log  # steps 24 .. 86: exactly the timed periods (Rig.finish flushes the last pair)
                assert len(fem.ms_log) <= 24 or len(ms) == steps, (len(fem.ms_log), steps)
                e["fem_ms_mean"] = round(sum(ms) / max(len(ms), 1), 3)
                e["fem_ms_min_max"] = [round(min(ms), 3), round(max(ms), 3)] if ms else None
                si = fem.sim.check_step(raise_on_penetration=False)
                e["fem_last_step"] = {"newton_iters_mean": round(float(si["newton_iters"].mean()), 2), "newton_iters_max": int(si["newton_iters"].max()),
                                      "pcg_iters_per_newton_mean": round(float((si["pcg_iters"] / np.maximum(si["newton_iters"], 1)).mean()), 1),
                                      "envs_flagged_penetration": int(len(si["penetrating_envs"])),
                                      "envs_flagged_line_search": int(len(si["line_search_failed_envs"]))}
                # the scene caps Newton at FemGelpad.max_newton_iter where the reference's default is 1024 (uipc_sim.py Newton.max_iter): the
                # measured rate only stands if NO env of NO logged step ran into the cap
                e["newton_cap"] = int(fem.max_newton_iter)
                e["newton_iters_max_over_period"] = int(fem.iters_max) if fem.iters_max is not None else None
                e["newton_cap_hit"] = bool(e["newton_iters_max_over_period"] is not None and e["newton_iters_max_over_period"] >= fem.max_newton_iter)
                assert cap_may_bind or not e["newton_cap_hit"], \
                    f"an env ran into the Newton cap of {fem.max_newton_iter} iterations: the FEM rate would be measured on truncated solves"
                tot = (fem.info_sum - base[0]).cpu().numpy()
                e["fem_period"] = {"steps": steps, "newton_iters_per_step_mean": round(float(tot[0]) / steps, 2),
                                   "pcg_iters_per_newton_mean": round(float(tot[3]) / max(float(tot[0]), 1e-9), 1),
                                   "note": "means over envs and over the timed window = three periods of the indenter's motion (21 steps each: about half "
                                           "pressing at ~1 ms per step, half following the retreating indenter at 3-15 ms)"}
                if hasattr(fem, "ind") and fem.sim.newton_kernel_resident:  # (the analytic-indenter scenes on the CU-resident Newton kernel: its roofline)
                    e["fem"] = fem_roofline(fem, (sum(ms) * steps / max(len(ms), 1), float(tot[0]), float(tot[3])))
                elif hasattr(fem, "ind"):
                    e["fem_kernel"] = "fem_newton_kernel (streaming form: one launch per Newton iteration; x, p, H.p accumulators in LDS)"
                else:
                    e["fem_kernel"] = "fem_ball_newton_kernel (csrc/fem_ball.h: one launch per time step)"
                    e["envs_flagged_overflow"] = int(len(si.get("pair_list_overflow_envs", [])))
            out.append(e)
            del rig
            torch.cuda.empty_cache()
        except Exception as ex:  # a sweep entry must not take the headline line down with it
            out.append({"key": key, "workload": label, "error": f"{type(ex).__name__}: {ex}"[:300]})

    E = args.envs_per_gpu
    run("c3_separate", f"C3 with the two sensors updated one by one (no GelSightSensorGroup): {E} envs x 2 sensors", E, 240, 320, 2, True, group=False)
    run("c3_no_gather", f"C3 without the observation gather / pack: {E} envs x 2 sensors, RGB 320x240 + FOTS markers", E, 240, 320, 2, True, gather="none")
    if not args.sensor_streams:
        run("c3_sensor_streams", f"C3 with one HIP stream per sensor: {E} envs x 2 sensors", E, 240, 320, 2, True, sensor_streams=True)
    run("c3_dense", f"C3 dense contact (data-independent floor): {E} envs x 2 sensors, RGB 320x240 + FOTS markers", E, 240, 320, 2, True, data="dense")
    run("ref_scene", "reference benchmark scene: 1024 envs x 1 sensor, camera 320x240 -> Taxim RGB 640x480 + FOTS 9x11 markers", 1024, 480, 640, 1, True,
        steps=max(5, args.sweep_steps // 3), count_in_contact=True, cam_res=(320, 240), clip=(0.024, 0.034), grid=(9, 11))
    run("c2", "C2: 256 envs x 1 sensor, Taxim RGB 320x240 (BASELINE configs[1])", 256, 240, 320, 1, False)
    run("c2_markers", "C2 + FOTS markers: 256 envs x 1 sensor, RGB 320x240 + markers", 256, 240, 320, 1, True)
    run("shard512", "512-env shard of the 4096-env / 8-GPU target: 512 envs x 1 sensor, RGB 320x240 + FOTS markers", 512, 240, 320, 1, True)
    run("c4", "C4 per-GPU shard: 512 envs, RGB 320x240 + FEM-driven markers + gelpad FEM step (1920 tets / env), FEM on a side stream",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev, max_newton_iter=NEWTON_CAP, side_stream=True))
    run("c4_one_stream", "C4 per-GPU shard on ONE stream (FEM step, then the sensor update)",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev, max_newton_iter=NEWTON_CAP))
    run("c4_dhat5e4", "C4 per-GPU shard with the reference scenes' contact zone d_hat = 5e-4 (ball_rolling_uipc.py:71-75)",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev, max_newton_iter=NEWTON_CAP, side_stream=True, d_hat=5e-4))
    run("c4_lag_capped", "C4 per-GPU shard with the opt-in reaction-capped friction lag (cfg.contact.friction_lag = 'capped') instead of IPC's previous-configuration lag (the default since round 6)",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev, max_newton_iter=NEWTON_CAP, side_stream=True, friction_lag="capped"))
    run("c4_rolling", "C4 shard, rolling contact: the indenter stays on the pad and slides, friction on",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev, motion="rolling", max_newton_iter=NEWTON_CAP, side_stream=True))
    # the reference's own UIPC scene (ball_rolling_uipc.py:71-125): free affine-body ball on the ground under the pad, pairs both ways, d_hat 5e-4.
    # with friction on every contact (the cfg's default contact model); stepped by csrc/fem_ball.h (one launch per time step)
    run("c4_ball", "C4 per-GPU shard on the reference's UIPC scene: 512 envs, RGB 320x240 + FEM-driven markers + gelpad FEM step against a FREE affine-body ball on the ground (point-triangle pairs both ways + edge-edge pairs + friction, d_hat 5e-4)",
        512, 240, 320, 1, False, fem=lambda: FemBallScene(512, dev, max_newton_iter=NEWTON_CAP, side_stream=True))
    run("c4_ball4096", "the reference's UIPC scene with the whole 4096-env job on ONE GPU (16 envs per CU instead of 2: the launch no longer ends with its slowest envs)",
        4096, 240, 320, 1, False, fem=lambda: FemBallScene(4096, dev, max_newton_iter=NEWTON_CAP, side_stream=True))
    run("c4_pad715", "C4 per-GPU shard with a pad the reference's mesher could just as well produce - 715 vertices / 2880 tets: beyond a CU's LDS, the streaming Newton kernel",
        512, 240, 320, 1, False, fem=lambda: FemGelpad(512, dev, max_newton_iter=NEWTON_CAP, side_stream=True, mesh=(10, 12, 4)))
    run("c5", "C5 per-GPU shard: 1024 envs, RGB 640x480 + FEM-driven markers (gelpad FEM step included) (BASELINE configs[4] / 8)",
        1024, 480, 640, 1, False, fem=lambda: FemGelpad(1024, dev, max_newton_iter=NEWTON_CAP, side_stream=True))
    # (with the coarse correction in M^-1 the reference's PCG test - 1e-3 on r.z - can pass after ONE iteration on this rod, whose coarse modes are
    #  nearly free: states then follow the tightly solved ones within the accumulated Newton tolerance.  The second entry solves to the threshold
    #  rounds 1-4 used - 1e-6 on r.z - which is also what the block-Jacobi streaming entry needs to mean the same accuracy.)
    if only is None or "axle" in only:
        out.append(fem_axle_entry(dev, key="axle"))
    if only is None or "axle_tol1e-6" in only:
        out.append(fem_axle_entry(dev, tol_rate=1e-6, key="axle_tol1e-6"))
    if only is None or "axle_streaming" in only:
        out.append(fem_axle_entry(dev, steps=6, streaming=True, tol_rate=1e-6, key="axle_streaming"))
    if only is None or "axle_streaming_2level" in only:
        out.append(fem_axle_entry(dev, steps=6, streaming=True, tol_rate=1e-6, key="axle_streaming_2level", two_level=True))
    run("c5_optical", "C5 optical part only: 1024 envs, RGB 640x480", 1024, 480, 640, 1, False, steps=max(5, args.sweep_steps // 3))
    return out


# What each sweep entry is, at length (goes to the details side file, never to the stdout line).
SWEEP_NOTES = {
    "c3_separate": "`--no-group`: two independent GelSightSensor.update() calls per step, each one launch sequence over its envs (rounds 1-4)",
    "c3_no_gather": "`--gather none`: the headline job without the 32x32 observation pack / all-gather",
    "c3_sensor_streams": "`--sensor-streams`: the left / right finger sensors of an env are independent objects; the drain and the small kernels of one "
                         "sensor's update overlap the other's.  NOT the default, because kernels that overlap have no duration of their own for the "
                         "roofline leg and a profile of the command to agree on",
    "c3_dense": "a wavy plate over the whole sensor: every frame, row and nearly every pixel in contact, so no zero band is skipped, no wave is flat "
                "and every table record is gathered",
    "ref_scene": "envs/ball_rolling_physx_rigid.py:161-199: camera 320x240 with clip (0.024, 0.034) up-sampled to the 640x480 tactile image; "
                 "frames_per_s_in_contact_only applies the reference's counting rule (run_ball_rolling_experiment.py:238-244)",
    "c4": "the FEM step runs on a HIP stream of its own, the sensor's optical pipeline overlaps its straggler tail and the FEM-driven markers wait "
          "for its event (FemGelpad side_stream; `fem_ms_*` then include the contention with the optical kernels)",
    "c4_one_stream": "A/B of the side stream; `fem_ms_*` are the FEM step alone",
    "c4_dhat5e4": "half the barrier width of UipcSimCfg's default (uipc_sim.py:103-124): what both of the reference's UIPC scenes set",
    "c4_rolling": "like the ball of the reference's ball-rolling scene (depth 0.3-0.8 of the maximum, sliding +-0.5 mm sideways); whenever the indenter "
                  "RETREATS the pad follows it up the barrier in damped Newton steps (every env runs to convergence: the cap is asserted never to bind)",
    "c4_ball": "every env converges at the reference's default tolerances (asserted: no env at the Newton cap, no flag); with 512 envs on 256 CUs the launch "
               "lasts as long as its slowest envs (6-9 Newton iterations where the mean is 1.4: envs whose ball has just been touched off-centre and turns)",
    "c4_ball4096": "the same scene, eight times the envs: what one GPU does with the north star's whole job; per-env cost = mean work, not the stragglers'",
    "axle": "SURVEY 8(d)'s ~2k-tet fixture simple_axle.msh (593 vertices / 2003 tets, scaled to 25.8 x 3 x 3 mm), ends attached, a sphere pressing "
            "on through the IPC barrier: the CU-resident Newton kernel, 768 threads per env (friction, coarse correction, chains: the defaults)",
    "axle_tol1e-6": "the same with the PCG threshold of rounds 1-4 (1e-6 on r.z)",
    "axle_streaming": "the streaming Newton kernel (deterministic switch; block Jacobi, friction off as in rounds 3-4), PCG threshold 1e-6",
    "axle_streaming_2level": "the same with the coarse correction on the bounding-box grid (round 6: the streaming kernel's two-level preconditioner)",
}


NEWTON_CAP = 64  # Newton iterations a FEM scene of the sweep may take per step (reference default 1024, uipc_sim.py Newton.max_iter):
                 # high enough that no env of no step reaches it - asserted, a truncated solve would flatter the rate


AXLE_NEWTON_CAP = 200  # (the bent axle's iterations in PSD-safe mode converge linearly: 50 in the worst env and step measured)


def fem_axle_entry(dev, B=512, steps=12, streaming=False, tol_rate=None, key="axle", two_level=False):
    """SURVEY section 8(d)'s ~2k-tet fixture simple_axle.msh (593 vertices / 2 003 tets) stepped with sphere contact - FEM only, env steps
    per second.  Default: the 768-thread variant of the CU-resident Newton kernel with everything the gelpad scene uses (friction,
    coarse correction on the bounding-box grid, the chains found in the mesh).  streaming=True: the streaming Newton kernel (what
    the deterministic switch selects for a mesh of more than 512 vertices; block Jacobi; run with friction off, as in rounds 3-4)."""
    name = "the streaming Newton kernel (deterministic switch; block Jacobi, friction off)" if streaming else \
           "the CU-resident Newton kernel, 768 threads per env (friction, coarse correction, chains: the defaults)"
    try:
        from tacex_amd.uipc import UipcObject, UipcObjectCfg, UipcSim, UipcSimCfg

        g = np.load(REPO / "tests" / "golden" / "fem_meshes.npz")
        P = (g["simple_axle_points"] - g["simple_axle_points"].min(0)) * 0.01
        T = g["simple_axle_tets"]
        cfg = UipcSimCfg(device=dev)
        cfg.newton.velocity_tol = 2e-3  # 20 um per step: the default (0.5 mm, uipc_sim.py:62-66) is a sixth of this rod's thickness
        cfg.contact.friction_lag = "capped"  # the documented setting for slender bodies (UipcSimCfg.Contact.friction_lag); gelpad scenes run "ipc"
        if tol_rate is not None:
            cfg.linear_system.tol_rate = tol_rate
        if streaming:
            cfg.linear_system.coarse_grid, cfg.linear_system.vertex_chains = ("auto" if two_level else None), None  # (round 6: the streaming kernel takes the coarse space)
            cfg.linear_system.deterministic = True
            cfg.contact.enable_friction = False  # (the entry's workload since round 3; the streaming kernel has friction since round 5)
        sim = UipcSim(cfg, num_envs=B)
        UipcObject(UipcObjectCfg(mesh_points=P, mesh_tets=T), sim)
        sim.setup_sim(constraint_strength_ratio=1000.0)
        ends = np.where((P[:, 0] < 0.002) | (P[:, 0] > P[:, 0].max() - 0.002))[0]
        sim.set_constraints(ends, torch.from_numpy(np.repeat(P[None, ends], B, 0)).to(dev))
        ind = torch.zeros((B, 8), dtype=torch.float64, device=dev)
        ind[:, 0], ind[:, 1], ind[:, 2], ind[:, 4] = 1.0, P[:, 0].max() / 2, P[:, 1].max() / 2, 0.004
        ind[:, 3] = P[:, 2].max() + 0.004 + 0.0009
        sim.set_contact_indenters(ind)
        ind = sim.contact_indenters
        depth = torch.linspace(0.2, 0.4, B, device=dev, dtype=torch.float64)
        its = torch.zeros((), dtype=torch.float64, device=dev)
        flagged = torch.zeros((), dtype=torch.float64, device=dev)

        def step(i):
            ind[:, 3] -= depth * sim.contact_gaps().amin(1)  # press on by a fraction of the gap ...
            if not streaming:
                ind[:, 1] += 2e-5 * (1 if (i // 4) % 2 == 0 else -1)  # ... and slide back and forth (friction acts)
            sim.step(max_newton_iter=AXLE_NEWTON_CAP)

        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # wall time of each third of the steps (a runtime stall inside one of them shows: see the C4 entries).  12 steps, not more: the
        # scene presses on by a fraction of the gap EVERY step - past step ~15 the rod is bent far enough that steps take 50-140 Newton
        # iterations at this tolerance (36 steps: thirds of 16 / 1 365 / 6 283 ms, profiles/r05_experiments.md section 16)
        thirds = []
        for w in range(3):
            tw = time.perf_counter()
            for i in range(w * (steps // 3), steps if w == 2 else (w + 1) * (steps // 3)):
                step(3 + i)
                its = torch.maximum(its, sim.step_info[:, 0].max())  # (the streaming path accumulates its per-launch counts in the same row)
                flagged = torch.maximum(flagged, (sim.step_info[:, 2].to(torch.int64) & 3).max().to(torch.float64))  # penetration / failed line search
            torch.cuda.synchronize()
            thirds.append(round((time.perf_counter() - tw) * 1e3, 2))
        el = time.perf_counter() - t0
        finite, gap = bool(torch.isfinite(sim.x).all()), float(sim.contact_gaps().amin())
        assert finite and gap > 0.0, f"finite {finite}, smallest gap {gap}"
        assert float(its) < AXLE_NEWTON_CAP, f"an env ran into the Newton cap of {AXLE_NEWTON_CAP}"
        return {"key": key, "workload": f"FEM only: {B} envs x simple_axle.msh, sphere contact: {name}",
                "envs": B, "steps": steps, "ms_per_step": round(el / steps * 1e3, 3), "env_steps_per_s": round(B * steps / el, 1), "thirds_ms": thirds,
                "newton_iters_max": int(its),
                "newton_cap": AXLE_NEWTON_CAP, "pcg_tol_rate": cfg.linear_system.tol_rate, "failure_flags_max": int(flagged), "velocity_tol": 2e-3}
    except Exception as ex:
        return {"key": key, "workload": f"FEM only: simple_axle.msh on {name}", "error": f"{type(ex).__name__}: {ex}"[:300]}


SWEEPS_PER_NEWTON = 2.0  # tet sweeps of fem_newton_lds_kernel per Newton iteration besides the PCG's (see fem_roofline)


def fem_roofline(fem, period=None):
    """What bounds the FEM step.  `period` = (total FEM ms, Newton iterations per env, PCG iterations per env) summed over the timed
    period of the scene: the roofline figures are taken over THAT (the scene's own mix of regimes); without it a single Newton
    iteration at the scene's current state is timed.  `UipcSim.step` runs ONE kernel (fem_newton_lds_kernel: the whole Newton loop with the env's
    state on the CU), so ITS roof is the one that matters: f64 vector throughput and LDS bandwidth of the CU the env sits on, not
    HBM (inside the PCG loop only the mesh constants are read, and those are shared by all envs and stay in L2).  Work per PCG
    iteration and env, counted from the kernel source (csrc/fem_kernels.hip, the `sweep` of the H.p product):
      per tet     F and dF from LDS x / p (2 x 27 FMA), tet state (cofactor 18 mul + 9 sub, Ic 9, J 3, coefficients ~12 incl. 2 div),
                  apply_dP (2 x 9 dot + 6 cross = 36 mul + 18 sub + 9 add, 4 x 9 axpy) and 12 row dots (36 FMA + 12 mul): ~300 f64
                  operations ~= 480 flop; LDS: 24 doubles read (x, p of 4 vertices), 12 written (rows)
      per vertex  ~24 incident tets x 3 doubles gathered from the LDS window, 3x3 block solve, CG updates: ~120 flop, 75 LDS doubles
    `element_terms` (fem_element_terms_kernel, HBM-bound assembled output) is NOT on the step path; it is timed for reference and
    its read bytes exclude the mesh constants (they are per mesh, not per env)."""
    sim = fem.sim
    B, T, V = fem.B, fem.num_tets, fem.num_verts

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

    if period is None:
        ms_nw = timeit(newton, 3)
        st = sim.stats.cpu().numpy()
        pcg = float(st[:, 3].mean())
        sim.x.copy_(x0)
        n_newton = 1.0
    else:
        ms_nw, n_newton, pcg = period
    # per env and PCG iteration (see the docstring).  Besides the PCG's H.p sweeps a Newton iteration runs SWEEPS_PER_NEWTON tet sweeps: the
    # gradient sweep and the line search's candidate energy (round 5; rounds 3-4 counted 4: the block assembly - now a kernel of its own - and
    # the line search's E(x) - now a by-product of the gradient sweep - were sweeps of this kernel then).  Both are priced like an H.p sweep,
    # which overstates them (no rows, no apply_dP): the f64 fraction is an upper estimate.
    flop_it = 480 * T + 120 * V
    lds_it = (36 * T + 75 * V) * 8
    sweeps = pcg + SWEEPS_PER_NEWTON * n_newton
    tf = flop_it * sweeps * B / (ms_nw * 1e-3) / 1e12
    lds_tbs = lds_it * sweeps * B / (ms_nw * 1e-3) / 1e12
    el_bytes = (12 * 8 + 8 + 96 + 1152) * B * T  # per env and tet: 4 vertices x 3 doubles read; energy + gradient + 12x12 Hessian written
    return {
        "newton_iteration": {"kernel": "fem_newton_lds_kernel (on the step path: 97 % of the FEM time)", "bound": "f64 VALU / LDS of one CU per env",
                             "ms": round(ms_nw, 3), "newton_iterations": round(n_newton, 2), "pcg_iterations": round(pcg, 1),
                             "window": "one Newton iteration at the current state" if period is None else "the timed period of the scene (sums per env)",
                             "us_per_sweep": round(ms_nw * 1e3 / max(sweeps, 1), 2),
                             "f64_flop_per_env_and_pcg_iteration": flop_it, "achieved_f64": round(tf, 2), "peak_f64": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / F64_PEAK_TFLOPS, 4),
                             "lds_bytes_per_env_and_pcg_iteration": lds_it, "achieved_lds": round(lds_tbs, 2), "peak_lds": LDS_PEAK_TBS,
                             "lds_unit": "TB/s", "lds_frac": round(lds_tbs / LDS_PEAK_TBS, 4),
                             "envs_per_cu": round(B / 256, 2),
                             "note": "one workgroup (512 threads, 2 waves/SIMD, 120-147 KB LDS) per env and CU; 256 CUs take 256 envs at a time; "
                                     "no HBM traffic inside the PCG loop (mesh constants in L2), so neither the HBM roof nor SURVEY 8(d)'s "
                                     "304 B/tet matrix-free figure binds; latency of the ~20 barriers and dependent LDS / L1 round trips per PCG "
                                     "iteration at 2 waves/SIMD is what the kernel waits on (section clock: profiles/r03_experiments.md section 8)"},
        "element_terms": {"kernel": "fem_element_terms_kernel (NOT on the step path; assembled-Hessian entry point of the C ABI)", "bound": "hbm",
                          "ms": round(ms_el, 4), "algo_bytes": el_bytes, "achieved": round(el_bytes / (ms_el * 1e-3) / 1e9, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(el_bytes / (ms_el * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "per env and tet: 96 B of vertex positions read (mesh constants are per mesh and stay in L2) + 8 + 96 + 1152 B written (fp64)"},
    }


def fem_roofline_entry(sw):
    """`roofline.fem`: the Newton kernel of the C4 shard against BOTH roofs north_star names for it - HBM (bytes per dispatch from the
    PMC passes of `scripts/fem_bench.py`, profiles/pmc_traffic_r0N_fem.json, over the kernel's mean duration in the kernel-trace run of
    the same command) and, live from the C4 sweep entry, the f64 vector rate and LDS rate of the CU an env sits on."""
    c4 = next((e for e in (sw or []) if e.get("key") == "c4" and "fem" in e), None)
    out = {"kernel": "fem_newton_lds_kernel<MESH = false, ATOM = true> (one dispatch = the whole Newton loop of a time step for all envs of the shard)"}
    for cand in ("pmc_traffic_r06_fem.json", "pmc_traffic_r05_fem.json", "pmc_traffic_r04_fem.json"):
        try:
            j = json.loads((REPO / "profiles" / cand).read_text())
            k = next(v for n, v in j["kernels"].items() if "fem_newton_lds_kernel<false" in n)  # (<MESH = false, ATOM = ...>)
            out.update({"hbm_bytes_per_dispatch": k["hbm_bytes_per_dispatch"], "mean_us_per_dispatch": k.get("mean_us_per_dispatch"),
                        "hbm_achieved": k.get("hbm_GBps"), "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": k.get("hbm_frac_of_8TBps"),
                        "hbm_source": f"profiles/{cand} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE + --kernel-trace --stats of scripts/fem_bench.py)",
                        "hbm_measured_at": j.get("measured_at_commit")})
            asm = next((v for n, v in j["kernels"].items() if "fem_assemble" in n), None)
            if asm is not None:
                out["assembly"] = {"kernel": "fem_assemble_blocks_kernel", "hbm_GBps": asm.get("hbm_GBps"), "hbm_frac": asm.get("hbm_frac_of_8TBps"),
                                   "mean_us_per_dispatch": asm.get("mean_us_per_dispatch")}
            break
        except Exception:
            out["hbm_achieved"] = None
    if c4 is not None:
        ni = c4["fem"]["newton_iteration"]
        per = c4.get("fem_period", {})
        out.update({"f64_achieved": ni["achieved_f64"], "f64_peak": ni["peak_f64"], "f64_unit": "TFLOP/s", "f64_frac": ni["frac"],
                    "lds_frac": ni["lds_frac"], "us_per_sweep": ni["us_per_sweep"], "window": ni["window"],
                    "sweeps_per_step": round(ni["pcg_iterations"] / max(per.get("steps", 1), 1) + SWEEPS_PER_NEWTON * ni["newton_iterations"] / max(per.get("steps", 1), 1), 2),
                    "newton_iters_per_step": per.get("newton_iters_per_step_mean"), "pcg_iters_per_newton": per.get("pcg_iters_per_newton_mean"),
                    "pcg_stop": PCG_STOP_RULE,
                    "matrix_free_bytes_per_tet_iteration_survey": 304,
                    "note": "the env's state lives on its CU (LDS + registers); inside the PCG loop only the mesh constants are read (shared "
                            "by all envs, L2), so the HBM roof does not bind this kernel"})
    return out if (c4 is not None or out.get("hbm_achieved") is not None) else None


PCG_STOP_RULE = "r.z <= tol_rate * r0.z0 (tol_rate 1e-3, uipc_sim.py linear_system default; rule restated from libuipc LinearPCG, source absent: unpinned)"


# ------------------------------------------------------------------------------------------------------------------------------
# The stdout line.  The driver keeps the LAST 8 KB of stdout: round 4's 20 KB line lost its head (BENCH_r04.json parsed: null).
# Everything that is not a headline number goes to the details side file; tests/test_bench_contract.py builds a line through
# compact_line() from a full synthetic sweep and bounds its length.
# ------------------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 4096  # bytes of the stdout line (asserted in emit(); the contract test allows 6000)

_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_avg_ms", "frames_per_launch",
              "algorithmic_bytes_per_launch", "frac_own_bytes", "pipeline_frac", "valu_frac")
_FEM_KEYS = ("hbm_frac", "hbm_achieved", "f64_frac", "lds_frac", "us_per_sweep", "sweeps_per_step", "newton_iters_per_step", "pcg_iters_per_newton")
_CPU_KEYS = ("value", "unit", "cores", "kind", "logical_cores", "physical_cores")
_SWEEP_SCALARS = {"c3_separate": "value_c3_separate", "c2": "value_c2", "c4": "value_c4", "c5": "value_c5", "c3_dense": "value_dense_contact", "c3_no_gather": "value_no_gather",
                  "c3_sensor_streams": "value_sensor_streams", "c4_rolling": "value_c4_rolling", "c4_lag_capped": "value_c4_lag_capped", "c4_dhat5e4": "value_c4_dhat5e4",
                  "c5_optical": "value_c5_optical", "shard512": "value_shard512", "c4_ball": "value_c4_ball", "c4_ball4096": "value_c4_ball4096",
                  "c4_pad715": "value_c4_pad715"}


def compact_line(full: dict, details_path: str | None) -> dict:
    """The ONE stdout line: the driver contract's keys + compact `roofline` / `cpu_baseline` + one scalar per sweep entry."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full[k] for k in keep}
    cfg = full["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "envs_per_gpu", "sensors_per_env", "frames_per_step", "resolution", "markers", "sensors", "gather", "arch")
                      if k in cfg}
    for e in cfg.get("sweep") or []:
        name = _SWEEP_SCALARS.get(e.get("key"))
        if name is not None:
            line[name] = e.get("frames_per_s")  # None = the entry failed (its error is in the details file)
    errs = [e["key"] for e in cfg.get("sweep") or [] if "error" in e]
    if errs:
        line["sweep_errors"] = errs
    r = full.get("roofline")
    if r is not None:
        line["roofline"] = {k: r[k] for k in _ROOF_KEYS if k in r}
        if r.get("fem"):
            line["roofline"]["fem"] = {k: r["fem"][k] for k in _FEM_KEYS if r["fem"].get(k) is not None}
            line["roofline"]["fem"]["pcg_stop"] = "r.z <= tol_rate*r0.z0"
    c = full.get("cpu_baseline")
    if c is not None:
        line["cpu_baseline"] = {k: c[k] for k in _CPU_KEYS if k in c}
        line["cpu_baseline"]["sample"] = c["sample"][:200]
    nd = full.get("node4096")
    if nd is not None:
        for k in ("value_node4096", "value_node4096_no_gather", "value_node4096_fem", "value_node4096_fem_no_gather", "strong_scaling_base"):
            line[k] = nd.get(k)
    m = full.get("multi_gpu")
    if m is not None:
        line["multi_gpu"] = {k: m[k] for k in ("backend", "world_size", "per_rank_ms_per_step", "value_no_gather", "ms_per_step_no_gather", "launcher")
                             if k in m}
        line["multi_gpu"]["distinct_devices"] = len(set(m.get("rank_devices", [])))
    if details_path:
        line["details"] = details_path
    return line


def emit(full: dict, details_path: str | None) -> str:
    """Writes the details side file (best effort) and returns the compact stdout line."""
    if details_path:
        try:
            dp = Path(details_path)
            dp.parent.mkdir(parents=True, exist_ok=True)
            full = dict(full, sweep_notes=SWEEP_NOTES)
            dp.write_text(json.dumps(full, indent=1))
        except OSError as ex:
            log(f"details file not written: {ex}")
            details_path = None
    if details_path:
        try:
            details_path = str(Path(details_path).resolve().relative_to(REPO))
        except ValueError:
            pass
    text = json.dumps(compact_line(full, details_path), separators=(",", ":"))
    assert len(text) <= LINE_BUDGET, f"stdout line of {len(text)} bytes (budget {LINE_BUDGET}): move the new field to the details file"
    return text


def launch_children(args, argv) -> int:
    """`python bench.py --gpus N` with no rendezvous in the environment: this process - which has made NO GPU call - starts N fresh
    children of this same script, one per GPU, with the torch.distributed.run environment contract (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*), waits for all of them and relays rank 0's line.  Nothing that touched the GPU is ever exec'ed or forked."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TACEX_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.stderr.write(out0)
        log(f"child exit codes {rcs}")
        return next(rc for rc in rcs if rc)
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    sys.stderr.write("".join(ln + "\n" for ln in out0.splitlines() if not ln.startswith("{")))
    print(lines[-1], flush=True)
    return 0


def dry_run_rank(args):
    """--cpu-dry-run: launcher, rendezvous (gloo), barrier-bracketed timing, max over ranks, the observation all-gather and the
    emitter, with a stand-in step on the CPU.  NOT a measurement (data = 'dry-run')."""
    from tacex_amd.env_shard import ObservationGather, init_from_env

    shard = init_from_env(args.envs_per_gpu * args.gpus, backend="gloo" if args.gpus > 1 else None)
    use_dist = dist.is_available() and dist.is_initialized()
    B = min(shard.num_local, 4)
    obs = ObservationGather({"rgb32_0": (32, 32, 3), "indent_0": (1,)}, B, shard.world_size, "cpu", dtypes={"rgb32_0": torch.uint8})

    def step(i):
        obs.pack_all({"rgb32_0": torch.full((B, 32, 32, 3), (i + shard.rank) % 251, dtype=torch.uint8), "indent_0": torch.full((B, 1), float(i))})
        obs.gather_async()

    for i in range(args.warmup):
        step(i)
    obs.wait()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    obs.wait()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    multi = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(shard.world_size)]
        dist.all_gather(every, t)
        elapsed = max(float(v) for v in every)
        multi = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank_devices": [f"cpu-{r}" for r in range(shard.world_size)],
                 "per_rank_ms_per_step": [round(float(v) / args.steps * 1e3, 4) for v in every],
                 "launcher": os.environ.get("TACEX_BENCH_LAUNCHER", "torch.distributed.run")}
        v = obs.views()
        assert v["indent_0"].shape[0] == B * shard.world_size and float(v["indent_0"][0, 0]) == args.warmup + args.steps - 1
    frames_per_step = args.envs_per_gpu * args.sensors * args.gpus
    full = headline_dict(args, frames_per_step * args.steps / elapsed, elapsed, not args.no_markers, "cpu (dry run)", None)
    full["data"] = "dry-run"
    if multi is not None:
        full["multi_gpu"] = multi
    if not args.no_node_leg:
        full["node4096"] = node4096_leg(args, shard, "cpu", use_dist, (lambda: dist.barrier()) if use_dist else (lambda: None), dry=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return emit(full, args.details_out) if shard.rank == 0 else None


NODE_ENVS = 4096  # the whole-node configuration `north_star`'s target is quoted on (BASELINE.json configs[3]: 4096 envs over the GPUs of one node)


class _DryRig:
    """Stand-in of Rig for --cpu-dry-run: the same timed() protocol around a CPU step that packs and (when asked) gathers a tiny observation."""

    def __init__(self, B, world, gather, collective=None):
        from tacex_amd.env_shard import ObservationGather

        self.B = min(B, 4)
        self.obs = ObservationGather({"rgb32_0": (32, 32, 3), "indent_0": (1,)}, self.B, world, "cpu", dtypes={"rgb32_0": torch.uint8},
                                     collective=collective) if gather else None

    def timed(self, steps, warmup, barrier=lambda: None, windows=1):
        def step(i):
            if self.obs is not None:
                self.obs.pack_all({"rgb32_0": torch.full((self.B, 32, 32, 3), i % 251, dtype=torch.uint8), "indent_0": torch.full((self.B, 1), float(i))})
                self.obs.gather_async()

        for i in range(warmup):
            step(i)
        if self.obs is not None:
            self.obs.wait()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        if self.obs is not None:
            self.obs.wait()
        barrier()
        return time.perf_counter() - t0


def node4096_leg(args, shard, dev, use_dist, barrier, dry=False):
    """the configuration the target names - 4096 envs over the WHOLE node, 4096 / N per GPU - on the N > 1 line (the
    headline weak-scales C3, 16 384 frames per step at N = 8, and never contained it).  Four rates, each the frames of all ranks over the
    MAX over ranks of the barrier-bracketed time: RGB 320x240 + FOTS markers with / without the observation all-gather, and C4-shaped
    (RGB + FEM-driven markers + the gelpad FEM step) with / without it.  `strong_scaling_base`: the same two jobs with all 4096 envs on ONE
    GPU (rank 0 alone, no collective, the other ranks wait at the barrier) - what the driver's per-N values of THESE keys divide by."""
    N = shard.world_size
    B = NODE_ENVS // N
    reduce_dev = "cpu" if dry else dev

    def make(kind, b, world, gather, collective=None):
        if dry:
            return _DryRig(b, world, gather, collective)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if kind == "fem":
            return Rig(b, 240, 320, 1, False, dev, world, seed=11 + shard.rank, gather="obs32" if gather else "none", obs_dtype=args.obs_dtype,
                       fem=FemGelpad(b, dev, max_newton_iter=NEWTON_CAP, side_stream=True), collective=collective)
        return Rig(b, 240, 320, 1, True, dev, world, seed=11 + shard.rank, gather="obs32" if gather else "none", obs_dtype=args.obs_dtype,
                   collective=collective)

    def rate(kind, b, world, gather, collective=None, sync=True):
        rig = make(kind, b, world, gather, collective)
        fem = kind == "fem" and not dry
        steps, warm = (63, 24) if fem else (max(5, args.steps // 2), max(2, args.warmup // 2))  # FEM: three periods of the indenter's motion (see sweep())
        el = rig.timed(steps, warm, barrier if sync else (lambda: None), windows=3 if fem else 1)
        if fem:
            assert rig.fem.iters_max is None or int(rig.fem.iters_max) < rig.fem.max_newton_iter, "an env ran into the Newton cap"
        if use_dist and sync:
            t = torch.tensor([el], device=reduce_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        del rig
        return round(b * world * steps / el, 1), round(el / steps * 1e3, 4)

    out = {"envs_total": B * N, "envs_per_gpu": B, "resolution": [320, 240], "sensors_per_env": 1}
    for kind, key in (("rgb", "node4096"), ("fem", "node4096_fem")):
        log(f"node leg: {key}: {B} envs per GPU x {N} GPUs")
        out[f"value_{key}"], out[f"ms_per_step_{key}"] = rate(kind, B, N, True)
        out[f"value_{key}_no_gather"], out[f"ms_per_step_{key}_no_gather"] = rate(kind, B, N, False)
    base = {}
    if N == 1:
        base = {"node4096": out["value_node4096"], "node4096_fem": out["value_node4096_fem"]}
    else:
        if shard.rank == 0:  # all 4096 envs on one GPU, no collective; the other ranks wait
            for kind, key in (("rgb", "node4096"), ("fem", "node4096_fem")):
                log(f"node leg: strong-scaling base {key}: {B * N} envs on rank 0 alone")
                base[key] = rate(kind, B * N, 1, True, collective=False, sync=False)[0]
        barrier()
    out["strong_scaling_base"] = base
    out["note"] = ("value_node4096*: 4096 envs x 1 GelSight Mini over the whole node (4096 / N per GPU), RGB 320x240 + FOTS markers; *_fem: RGB + FEM-driven "
                   "markers + gelpad FEM step (C4); each with and without the one observation all-gather per step; strong_scaling_base = the same jobs "
                   "with all 4096 envs on one GPU")
    return out



def headline_dict(args, value, elapsed, markers, arch, obs_bytes, sensor_streams_on=False, use_dist=False, grouped=False):
    W, H = args.width, args.height
    frames_per_step = args.envs_per_gpu * args.sensors * args.gpus
    return {
        "metric": "tactile_frames_per_sec", "value": round(value, 1), "unit": "frames/s", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.envs_per_gpu} envs x {args.sensors} GelSight Mini per GPU = {args.envs_per_gpu * args.sensors} frames/step/GPU, "
                        f"Taxim RGB {W}x{H}" + (" + FOTS markers (99)" if markers else "")
                        + ("; BASELINE configs[2] (C3)" if (args.envs_per_gpu, args.sensors, W, H, markers) == (1024, 2, 320, 240, True) else ""),
            "envs_per_gpu": args.envs_per_gpu, "sensors_per_env": args.sensors, "frames_per_step": frames_per_step,
            "resolution": [W, H], "markers": markers,
            "gather": "none" if obs_bytes is None else f"obs32 {args.obs_dtype} ({obs_bytes} B/rank, " + ("1 all_gather/step)" if (args.gpus > 1 or use_dist) else "N=1: no collective)"),
            "sensors": ("group" if grouped else "separate"),
            "two_sensor_batching": BATCHING_NOTE + (", evaluated as ONE GelSightSensorGroup: one launch sequence over the frames of both" if grouped else
                                                    ", each one launch sequence over its envs")
                                   + (", one HIP stream per sensor (joined before the observation is packed)" if sensor_streams_on else ""),
            "observation_gather": None if obs_bytes is None else {
                "payload": f"per sensor: 32x32x3 {args.obs_dtype} RGB (antialiased, produced in the render pass) + f32 indentation"
                           + (" + f32 markers (2,99,2)" if markers else ""),
                "bytes_per_rank": obs_bytes,
                "collective": "all_gather_into_tensor x1 per step" if (args.gpus > 1 or use_dist) else "none (N=1: the packed buffer is the observation)"},
            "background_frame": "synthetic f0 (real dataPack.npz absent from the reference checkout)",
            "arch": arch,
        },
    }


BATCHING_NOTE = "two independent GelSightSensor objects (gsmini_left / gsmini_right as factory_env_cfg.py:192-213)"


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.details_out is None:
        args.details_out = str(REPO / "gpurun_out" / f"bench_details_n{args.gpus}.json")
    # (TACEX_BENCH_FORCE_LAUNCH=1: take the self-launch path at N = 1 too - the GPU box has one device, and the launcher + RCCL
    #  rendezvous + relay must not meet real hardware for the first time on the 8-GPU node; tests/test_env_shard_gloo.py)
    if (args.gpus > 1 or os.environ.get("TACEX_BENCH_FORCE_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_children(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Native libraries write banners to the C stdout (RCCL prints its version block when the first communicator is made, and
    # libc only flushes it at exit - after our line; gloo prints its peer count).  The driver parses stdout for ONE JSON line:
    # everything before it goes to stderr instead (fd 1 -> fd 2 until the line is printed).
    import ctypes
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    def restore_stdout():
        try:
            ctypes.CDLL(None).fflush(None)  # native buffers (the RCCL banner) land on stderr
        except OSError:
            pass
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    if args.cpu_dry_run:
        text = dry_run_rank(args)
        restore_stdout()
        if text is not None:
            print(text, flush=True)
        return

    from tacex_amd import _lib
    from tacex_amd.env_shard import init_from_env

    H, W = args.height, args.width
    markers = not args.no_markers
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
              obs_dtype=args.obs_dtype, sensor_streams=args.sensor_streams, group=not args.no_group)
    elapsed = rig.timed(args.steps, args.warmup, barrier)
    log(f"headline timed: {elapsed / args.steps * 1e3:.3f} ms/step")
    multi = None
    if use_dist:
        # self-proving N > 1 line: every rank's own time, the devices behind the ranks (all-gathered over RCCL itself)
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.empty((shard.world_size,), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        props = torch.cuda.get_device_properties(shard.local_rank)
        ident = f"{props.name}|{getattr(props, 'uuid', '')}|pci {getattr(props, 'pci_bus_id', '?')}"
        blob = torch.zeros((96,), dtype=torch.uint8, device=dev)
        raw = ident.encode()[:96]
        blob[: len(raw)] = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
        blobs = torch.empty((shard.world_size * 96,), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(blobs, blob)
        multi = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                 "rank_devices": [bytes(blobs[r * 96:(r + 1) * 96].cpu().tolist()).rstrip(b"\0").decode(errors="replace")
                                  for r in range(shard.world_size)],
                 "per_rank_ms_per_step": [round(float(v) / args.steps * 1e3, 4) for v in every.cpu().tolist()],
                 "launcher": os.environ.get("TACEX_BENCH_LAUNCHER", "torch.distributed.run")}
        elapsed = float(every.max().item())
    frames_per_step = args.envs_per_gpu * args.sensors * args.gpus
    value = frames_per_step * args.steps / elapsed
    # the same job without the observation collection (frames/s with AND without the gather), N > 1 only: at N = 1
    # the sweep carries the `--gather none` entry
    if use_dist and args.gather != "none" and args.gpus > 1:
        rig2 = Rig(B, H, W, args.sensors, markers, dev, shard.world_size, seed=1 + shard.rank, gather="none", obs_dtype=args.obs_dtype,
                   sensor_streams=args.sensor_streams, group=not args.no_group)
        e2 = rig2.timed(max(5, args.steps // 4), 2, barrier)
        t2 = torch.tensor([e2], device=dev, dtype=torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        multi["value_no_gather"] = round(frames_per_step * max(5, args.steps // 4) / float(t2.item()), 1)
        multi["ms_per_step_no_gather"] = round(float(t2.item()) / max(5, args.steps // 4) * 1e3, 4)
        del rig2
        torch.cuda.empty_cache()

    roofline = None
    if not args.no_roofline and shard.rank == 0:
        roofline = roofline_leg(rig, markers)
    obs_bytes = None if rig.obs is None else rig.obs.payload_bytes()
    sensor_streams_on = bool(rig.streams)
    grouped = rig.grouped
    del rig
    torch.cuda.empty_cache()

    node = None
    if not args.no_node_leg:
        try:
            node = node4096_leg(args, shard, dev, use_dist, barrier)
        except Exception as ex:
            if use_dist:
                raise  # (a rank that fails alone would leave the others in a collective)
            node = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    sw = None
    if args.gpus == 1 and not args.no_sweep and shard.rank == 0 and not use_dist:
        sw = sweep(args, dev)

    cpu = None
    if not args.no_cpu_baseline and shard.rank == 0 and args.gpus == 1:
        log("cpu baseline leg")
        cpu = cpu_baseline(args.cpu_baseline_seconds)
    log("done")

    text = None
    if shard.rank == 0:
        full = headline_dict(args, value, elapsed, markers, _lib.require_gpu(shard.local_rank), obs_bytes, sensor_streams_on, use_dist, grouped)
        if multi is not None:
            full["multi_gpu"] = multi
        if node is not None:
            full["node4096"] = node
        if sw is not None:
            full["config"]["sweep"] = sw
        if roofline is not None:
            full["roofline"] = roofline
            fem_leg = fem_roofline_entry(sw)
            if fem_leg is not None:
                roofline["fem"] = fem_leg
        if cpu is not None:
            full["cpu_baseline"] = cpu
        text = emit(full, args.details_out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    restore_stdout()
    if text is not None:
        print(text, flush=True)


if __name__ == "__main__":
    main()
