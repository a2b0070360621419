#!/usr/bin/env python3
"""Benchmark of the tactile hot path through the drop-in boundary (GelSightSensor.update()).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one `sensor.update(dt, force_recompute=True)` over this rank's env shard: camera depth (already resident
in HBM) -> height map + indentation depth -> Taxim RGB 320x240 -> FOTS markers, then the low-resolution policy
observation (32x32x3 antialiased downsample + markers + indentation) is packed and collected with ONE all-gather
(RCCL over xGMI when N > 1).  Workload = BASELINE.json configs[1] (256 envs x 1 GelSight Mini per GPU, weak
scaling) with the FOTS markers the metric names.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
FP32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=256, help="env shard per GPU (weak scaling)")
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--no-markers", action="store_true", help="Taxim RGB only (skip the FOTS marker field)")
    ap.add_argument("--gather", choices=["obs32", "none"], default="obs32",
                    help="payload of the per-step observation all-gather (obs32 = 32x32x3 RGB + markers + indentation)")
    ap.add_argument("--obs-dtype", choices=["u8", "f32"], default="u8",
                    help="dtype of the 32x32x3 policy image in the gather payload (u8 = what a CNN policy consumes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    return ap.parse_args()


def build_sensor(num_envs, H, W, markers, device, obs_res=None, obs_dtype="float32"):
    from tacex_amd import GelSightSensor, GelSightSensorCfg
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.simulation_approaches.fots import FOTSMarkerSimulatorCfg
    from tacex_amd.simulation_approaches.gpu_taxim import TaximSimulatorCfg

    types = ["tactile_rgb", "height_map"] + (["marker_motion"] if markers else [])
    cfg = GelSightSensorCfg(
        num_envs=num_envs,
        sensor_camera_cfg=GelSightSensorCfg.SensorCameraCfg(resolution=(W, H), clipping_range=(0.024, 0.029)),
        data_types=types,
        optical_sim_cfg=TaximSimulatorCfg(calib_folder_path=str(CALIB_GELSIGHT_MINI), gelpad_height=0.0045,
                                          gelpad_to_camera_min_distance=0.024, with_shadow=False,
                                          tactile_img_res=(W, H), device=device, policy_obs_res=obs_res,
                                          policy_obs_dtype=obs_dtype),
        marker_motion_sim_cfg=FOTSMarkerSimulatorCfg(tactile_img_res=(W, H), device=device) if markers else None,
        device=device,
    )
    s = GelSightSensor(cfg)
    s.initialize()
    return s


def cpu_baseline(H, W, seconds):
    """Reference CPU path (FFT-faithful torch-CPU port, oracle/taxim_torch_cpu.py) on this box's host cores."""
    from oracle.taxim_torch_cpu import TaximTorchCpuPort
    from tacex_amd.calibration import CALIB_GELSIGHT_MINI
    from tacex_amd.utils.synthetic import synthetic_depth_maps

    B = 16
    port = TaximTorchCpuPort(CALIB_GELSIGHT_MINI, (H, W))
    hm, ind = synthetic_depth_maps(B, H, W, seed=1, device="cpu")
    # pick the intra-op thread count that serves this FFT-heavy path best on this host (all cores is usually
    # NOT the best: 256 threads on small FFTs oversubscribe badly); candidates are timed on one call each
    ncpu = os.cpu_count() or 1
    best_t, best = None, float("inf")
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        port.render_direct(hm[:4], ind[:4])  # warm-up
        t0 = time.perf_counter()
        port.render_direct(hm, ind)
        dt = time.perf_counter() - t0
        if dt < best:
            best_t, best = th, dt
    torch.set_num_threads(best_t)
    n, t0 = 0, time.perf_counter()
    while True:
        port.render_direct(hm, ind)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": round(B * n / el, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} calls x {B} frames {W}x{H}, Taxim RGB no-shadow (FFT blur, torch CPU), same synthetic depth maps (seed 1)"}


def main():
    args = parse()
    from tacex_amd import _lib
    from tacex_amd.env_shard import ObservationGather, init_from_env
    from tacex_amd.utils.synthetic import synthetic_depth_maps

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
    sensor = build_sensor(B, H, W, markers, dev, obs_res=(32, 32) if args.gather == "obs32" else None,
                          obs_dtype="uint8" if args.obs_dtype == "u8" else "float32")
    # synthetic camera depth (metres), already resident in HBM; a different seed per shard
    hm_mm, _ = synthetic_depth_maps(B, H, W, seed=1 + shard.rank, device=dev)
    depth_m = (hm_mm / 1000.0).contiguous()
    del hm_mm
    theta = torch.zeros(B, device=dev)
    sensor.set_camera_depth(depth_m)
    lib = _lib.load_library()

    obs = None
    if args.gather == "obs32":
        pieces = {"rgb32": (32, 32, 3), "indent": (1,)}
        if markers:
            pieces["markers"] = (2, 99, 2)
        obs = ObservationGather(pieces, B, shard.world_size, dev,
                                dtypes={"rgb32": torch.uint8} if args.obs_dtype == "u8" else None)

    def step(i: int):
        if markers:
            sensor.marker_motion_simulator.set_indenter_yaw(theta)
        sensor.update(dt=0.01, force_recompute=True)
        if obs is not None:
            out = sensor._data.output
            vals = {"rgb32": out["tactile_rgb_obs"],  # produced inside the render pass (fused into the tail kernel)
                    "indent": sensor.indentation_depth}
            if markers:
                vals["markers"] = out["marker_motion"]
            obs.pack_all(vals)
            obs.gather_async()  # overlaps the next step's rendering; ordered before the next pack / the final sync

    use_dist = dist.is_available() and dist.is_initialized()

    def barrier():
        if use_dist:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    if obs is not None:
        obs.wait()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if obs is not None:
        obs.wait()  # the last step's collective is part of the timed region
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_frames = args.envs_per_gpu * args.gpus * args.steps
    value = total_frames / elapsed

    # ---- roofline leg: per-stage hipEvent timing on the launch stream (rank 0, outside the timed region) ----
    roofline = None
    if not args.no_roofline and shard.rank == 0:
        taxim = sensor.optical_simulator._taxim
        taxim.set_profiling((H, W), True)
        for i in range(10):
            sensor.update(dt=0.01, force_recompute=True)
        torch.cuda.synchronize()
        prof = taxim.read_profile((H, W))
        taxim.set_profiling((H, W), False)
        N = H * W
        stages = {}
        for name, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            if name == "frame_min":
                bpf = 4 * N
            elif name.startswith("blur_l0"):
                bpf = 8 * N          # read height map, write level 0
            elif name.startswith("blur_"):
                bpf = 12 * N         # read previous level + height map (masked restore), write level
            elif name.startswith("tail"):
                # read level + height map, write RGB; FOTS gets the marker pixels + per-wave statistics from the same kernel
                # (a few KB per frame), so the full deformed-gel / mask frames (5 B/px) are only stored on the fallback path
                full_frames = markers and getattr(sensor.optical_simulator, "_fots_compact_version", -1) < 0
                bpf = (20 + (5 if full_frames else 0)) * N
            else:
                bpf = 16 * N         # shade: read deformed gel, write RGB
            avg = ms / cnt
            stages[name] = {"avg_ms": round(avg, 5), "algo_bytes_per_launch": bpf * B,
                            "GBps": round(bpf * B / (avg * 1e-3) / 1e9, 1)}
        dom = max(stages, key=lambda k: stages[k]["avg_ms"])
        taxim_ms = sum(s["avg_ms"] for s in stages.values())
        ach = stages[dom]["GBps"]
        pipeline_gbs = 16 * N * B / (taxim_ms * 1e-3) / 1e9
        flops_per_frame = 2 * sum(kw + kh for kw, kh in zip(taxim.context((H, W)).tables.ksize_w,
                                                              taxim.context((H, W)).tables.ksize_h)) * N + 100 * N
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "pipeline_achieved": round(pipeline_gbs, 1), "pipeline_frac": round(pipeline_gbs / HBM_PEAK_GBS, 4),
            "valu_frac": round(flops_per_frame * B / (taxim_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
            "stages": stages,
            "note": "achieved = algorithmic bytes of the dominant kernel per launch / its hipEvent-measured duration; "
                    "pipeline_* = 16 B/px compulsory bytes of the whole Taxim path / sum of its kernels; "
                    "valu_frac = algorithmic fp32 flops / 157.3 TFLOP/s (the separable blur is VALU-heavy)",
        }
        # HBM bytes of the dominant kernel per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected with
        # separate `rocprofv3 --pmc` runs and committed as profiles/pmc_traffic.json - counters cannot be read live here)
        pmc = REPO / "profiles" / "pmc_traffic.json"
        if pmc.exists() and (H, W) == (240, 320):
            try:
                per_frame = json.loads(pmc.read_text())["per_frame_bytes"].get(dom)
                if per_frame is not None:
                    roofline["traffic"] = int(per_frame * B)
                    roofline["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per-frame bytes x frames per launch)"
            except Exception:
                pass

    cpu = None
    if not args.no_cpu_baseline and shard.rank == 0 and args.gpus == 1:
        cpu = cpu_baseline(H, W, args.cpu_baseline_seconds)

    if shard.rank == 0:
        line = {
            "metric": "tactile_frames_per_sec", "value": round(value, 1), "unit": "frames/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.envs_per_gpu} envs x 1 GelSight Mini per GPU, Taxim RGB {W}x{H}"
                            + (" + FOTS markers (99)" if markers else "") + " via GelSightSensor.update(); BASELINE configs[1]"
                            + (" + markers" if markers else ""),
                "envs_per_gpu": args.envs_per_gpu, "resolution": [W, H], "markers": markers,
                "observation_gather": None if obs is None else {"payload": f"32x32x3 {args.obs_dtype} RGB (antialiased, produced in the render pass) + f32 indentation"
                                                                + (" + f32 markers (2,99,2)" if markers else ""),
                                                                "bytes_per_rank": obs.payload_bytes(),
                                                                "collective": "all_gather_into_tensor x1 per step" if (args.gpus > 1 or use_dist) else "none (N=1: the packed buffer is the observation)"},
                "background_frame": "synthetic f0 (real dataPack.npz absent from the reference checkout)",
                "arch": _lib.require_gpu(shard.local_rank),
            },
        }
        if roofline is not None:
            line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
