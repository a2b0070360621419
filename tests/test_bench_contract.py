"""The committed bench line (profiles/r04_bench_n1.json = stdout of `python bench.py` on the GPU box) follows the driver's
contract: the headline keys, the `roofline` and `cpu_baseline` objects, the workload naming - and agrees with the kernel stats
of the rocprofv3 run of the same command that sits beside it (same gpurun call, same box)."""
import csv
import json
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def test_committed_bench_line_follows_the_contract():
    d = json.loads((REPO / "profiles" / "r04_bench_n1.json").read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.loads((REPO / "BASELINE.json").read_text())
    assert "tactile frames/sec" in base["metric"] and d["metric"] == "tactile_frames_per_sec" and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert {"C2", "C4", "C5"} <= {s["workload"][:2] for s in d["config"]["sweep"]}
    # roofline.achieved = 16 B/px x the pixels of one launch of the dominant kernel / its measured duration
    W, H = d["config"]["resolution"]
    assert r["algorithmic_bytes_per_launch"] == 16 * W * H * r["frames_per_launch"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) <= 1e-3 * r["achieved"]
    assert r["frac_own_bytes"] >= r["frac"]
    # the FEM scenes run every env to convergence: the cap is reported and must not have been hit (the continuity entry says it caps)
    for e in d["config"]["sweep"]:
        if "newton_cap" in e and "truncated_solves" not in e and "env_steps_per_s" not in e:
            assert e["newton_iters_max_over_period"] < e["newton_cap"] and e["newton_cap_hit"] is False, e["workload"][:40]
    f = r["fem"]
    assert f["hbm_achieved"] is not None and 0 < f["hbm_frac"] < 1 and 0 < f["f64_frac"] < 1


def test_roofline_duration_agrees_with_the_rocprof_summary():
    d = json.loads((REPO / "profiles" / "r04_bench_n1.json").read_text())
    stage = d["roofline"]["stages"][d["roofline"]["kernel"]]
    rows = list(csv.DictReader(open(REPO / "profiles" / "r04_c3_kernel_stats.csv")))
    k = next(r for r in rows if "taxim_stream_kernel" in r["Name"])
    assert abs(float(k["AverageNs"]) * 1e-6 - stage["avg_ms"]) <= 0.05 * stage["avg_ms"]  # hipEvent vs profiler: within 5 %


def test_every_traffic_file_is_at_least_its_compulsory_bytes():
    """A per-frame HBM traffic figure below the compulsory 16 B/px means the counters were divided by the wrong frame count
    (round 3's 640x480 file: 1024 frames per tail launch assumed where the pass launches 256).  `scripts/make_pmc_traffic.py` now
    takes the frame count from the tail's own WRITE_SIZE and refuses such a result; this pins the committed files."""
    files = sorted((REPO / "profiles").glob("pmc_traffic_r*.json"))
    assert files
    checked = 0
    for f in files:
        j = json.loads(f.read_text())
        if "taxim_path_sum_per_frame" not in j:
            continue  # FEM files: per-dispatch figures, no per-frame path sum
        checked += 1
        assert j["taxim_path_sum_per_frame"] >= j["compulsory_per_frame_16B_per_px"], f.name
        W, H = j["resolution"]
        assert j["compulsory_per_frame_16B_per_px"] == 16 * H * W, f.name
        tail = j["per_frame_bytes"].get("tail_fused")
        if tail is not None:  # the tail alone writes 12 B/px of RGB and reads >= 4 B/px of the last band level
            assert tail >= 0.95 * 16 * H * W, (f.name, tail)
    assert checked >= 2
