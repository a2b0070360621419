"""The line `bench.py` prints follows the driver's contract - the headline keys, compact `roofline` and `cpu_baseline` objects,
the workload naming - stays within the driver's stdout tail whatever the sweep holds, and (the committed copy of a GPU-box run)
agrees with the kernel stats of the rocprofv3 run of the same command that sits beside it (same gpurun call, same box)."""
import csv
import json
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline")


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", REPO / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _synthetic_full(bench):
    """A worst-case full record: every sweep entry with paragraph-long strings, all FEM blocks, a long CPU thread sweep."""
    args = bench.parse(["--steps", "20", "--warmup", "5"])
    full = bench.headline_dict(args, 650000.0, 20 * 3.15e-3, True, "gfx950:sramecc+:xnack-", 6400 * 1024)
    long = "x" * 700
    fem = {"newton_iteration": {"kernel": long, "ms": 1.0, "newton_iterations": 23.0, "pcg_iterations": 80.0, "us_per_sweep": 88.0, "achieved_f64": 5.7,
                                "peak_f64": 78.6, "frac": 0.07, "lds_frac": 0.06, "note": long, "window": long},
           "element_terms": {"kernel": long, "note": long, "frac": 0.7}}
    sw = []
    for key in list(bench._SWEEP_SCALARS) + ["ref_scene", "c2_markers", "c4_one_stream", "axle", "axle_tol1e-6", "axle_streaming"]:
        e = {"key": key, "workload": long, "frames_per_step": 2048, "steps": 30, "ms_per_step": 3.1234, "frames_per_s": 612345.6, "note": long}
        if key.startswith(("c4", "c5")) and key != "c5_optical":
            e.update({"fem_ms_mean": 0.5, "fem_ms_min_max": [0.3, 3.0], "fem": fem, "newton_cap": 64, "newton_iters_max_over_period": 9, "newton_cap_hit": False,
                      "fem_period": {"steps": 21, "newton_iters_per_step_mean": 1.1, "pcg_iters_per_newton_mean": 4.2, "note": long}})
        if key.startswith("axle"):
            e["env_steps_per_s"] = 119000.0
        sw.append(e)
    sw.append({"key": "c5", "workload": long, "error": long[:300]})
    full["config"]["sweep"] = sw
    full["roofline"] = {"bound": "hbm", "kernel": "tail_fused", "achieved": 1620.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.2025, "traffic": 1300000000,
                        "frames_per_launch": 1024, "kernel_avg_ms": 0.777, "algorithmic_bytes_per_launch": 16 * 76800 * 1024, "frac_own_bytes": 0.25,
                        "pipeline_frac": 0.1, "valu_frac": 0.2, "stages": {f"stage{i}": {"avg_ms": 0.1, "note": long} for i in range(12)}, "note": long * 3,
                        "fem": bench.fem_roofline_entry(sw)}
    full["cpu_baseline"] = {"value": 206.3, "unit": "frames/s", "cores": 16, "logical_cores": 256, "physical_cores": 128, "kind": "port", "sample": long,
                            "runs": [{"note": long}] * 4, "thread_sweep_B16": [{"threads": t, "frames_per_s": 1.0} for t in (8, 16, 32, 64, 128)],
                            "protocol_all_cores": {"note": long}}
    full["node4096"] = {"envs_total": 4096, "envs_per_gpu": 512, "value_node4096": 5.1e6, "value_node4096_no_gather": 5.2e6, "value_node4096_fem": 3.0e6,
                        "value_node4096_fem_no_gather": 3.1e6, "strong_scaling_base": {"node4096": 7.0e5, "node4096_fem": 3.9e5}, "note": long}
    full["multi_gpu"] = {"backend": "nccl", "world_size": 8, "rank_devices": [f"AMD Instinct MI355X|{i}" for i in range(8)],
                         "per_rank_ms_per_step": [3.1] * 8, "value_no_gather": 1.0, "ms_per_step_no_gather": 1.0, "launcher": "self"}
    return full


def test_stdout_line_is_compact_whatever_the_sweep_holds(tmp_path):
    """BENCH_r04.json had `parsed: null`: the line was 20 KB and the driver keeps the last 8 KB of stdout.  The line bench.py prints is
    built by emit() -> compact_line(); with a worst-case full record it must stay within the budget and keep the contract's keys."""
    bench = _load_bench()
    full = _synthetic_full(bench)
    text = bench.emit(full, str(tmp_path / "details.json"))
    assert len(text) <= 6000 and len(text) <= bench.LINE_BUDGET and "\n" not in text
    d = json.loads(text)
    for k in HEAD_KEYS:
        assert k in d, k
    assert set(d["config"]) == {"workload", "envs_per_gpu", "sensors_per_env", "frames_per_step", "resolution", "markers", "sensors", "gather", "arch"}
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_avg_ms", "frames_per_launch", "fem"} <= set(d["roofline"])
    assert {"f64_frac", "us_per_sweep", "sweeps_per_step", "pcg_stop"} <= set(d["roofline"]["fem"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    for k in ("value_c2", "value_c4", "value_dense_contact", "value_no_gather", "value_c4_dhat5e4"):
        assert d[k] == 612345.6, k
    assert d["value_c5"] is None and d["sweep_errors"] == ["c5"]  # (the failed entry is last in the list: it wins the scalar)
    assert d["multi_gpu"]["distinct_devices"] == 8
    # the 4096-env whole-node point is on the line; the cherry-picked axle phase is not
    assert d["value_node4096"] == 5.1e6 and d["value_node4096_fem_no_gather"] == 3.1e6 and d["strong_scaling_base"] == {"node4096": 7.0e5, "node4096_fem": 3.9e5}
    assert "value_axle_env_steps" not in d
    # nothing is lost: the side file holds the full record
    det = json.loads((tmp_path / "details.json").read_text())
    assert len(det["config"]["sweep"]) == len(full["config"]["sweep"]) and "stages" in det["roofline"] and "sweep_notes" in det


def _committed(name):
    f = REPO / "profiles" / name
    if not f.exists():
        import pytest
        pytest.skip(f"{name} not committed yet")
    return json.loads(f.read_text())


def test_committed_bench_line_follows_the_contract():
    """profiles/r06_bench_n1.json = the stdout line of `python bench.py` on the GPU box; r06_bench_details_n1.json = its side file."""
    raw = (REPO / "profiles" / "r06_bench_n1.json")
    d = _committed("r06_bench_n1.json")
    assert len(raw.read_text().strip()) <= 6000
    for k in HEAD_KEYS:
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
    for k in ("value_c2", "value_c4", "value_c5"):
        assert d[k] and d[k] > 0, k
    # roofline.achieved = 16 B/px x the pixels of one launch of the dominant kernel / its measured duration
    W, H = d["config"]["resolution"]
    assert r["algorithmic_bytes_per_launch"] == 16 * W * H * r["frames_per_launch"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) <= 1e-3 * r["achieved"]
    f = r["fem"]
    assert 0 < f["f64_frac"] < 1 and f["us_per_sweep"] > 0 and f["sweeps_per_step"] > 0
    # the FEM scenes run every env to convergence: the cap is reported in the details file and must not have been hit
    det = _committed("r06_bench_details_n1.json")
    for e in det["config"]["sweep"]:
        if "newton_cap" in e and "env_steps_per_s" not in e:
            assert e["newton_iters_max_over_period"] < e["newton_cap"] and e["newton_cap_hit"] is False, e["key"]


def test_roofline_duration_agrees_with_the_rocprof_summary():
    d = _committed("r06_bench_n1.json")
    f = REPO / "profiles" / "r06_c3_kernel_stats.csv"
    if not f.exists():
        import pytest
        pytest.skip("r06 kernel stats not committed yet")
    rows = list(csv.DictReader(open(f)))
    k = next(r for r in rows if "taxim_stream_kernel" in r["Name"])
    avg = d["roofline"]["kernel_avg_ms"]
    assert abs(float(k["AverageNs"]) * 1e-6 - avg) <= 0.05 * avg  # hipEvent vs profiler: within 5 %


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
