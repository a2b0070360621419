"""Build profiles/pmc_traffic_rNN_fem.json: HBM bytes per dispatch of the FEM kernels from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE in SEPARATE runs of `python3 scripts/fem_bench.py`, scripts/gpu_round.sh <tag> pmcfem) and, from the kernel-trace --stats run
of the same command (gpu_round.sh <tag> proffem), their mean duration - so that every kernel gets its achieved HBM GB/s.
usage: python scripts/make_pmc_traffic_fem.py <pmcfem_fetch dir> <pmcfem_write dir> <proffem dir> <out.json> [commit]"""
import csv, glob, json, os, sys
from collections import defaultdict

fetch_dir, write_dir, stats_dir, out = sys.argv[1:5]
commit = sys.argv[5] if len(sys.argv) > 5 else "unknown"


def short(name):
    n = name.split("(")[0].strip()
    return n.replace("void ", "")


def collect(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "fem_" in row["Kernel_Name"]:
                k = short(row["Kernel_Name"])
                acc[k] += float(row["Counter_Value"]); cnt[k] += 1
    return acc, cnt


fa, fc = collect(fetch_dir, "FETCH_SIZE")
wa, wc = collect(write_dir, "WRITE_SIZE")
dur = {}
for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "fem_" in row["Name"]:
            dur[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))
res = {}
for k in sorted(set(fa) | set(wa)):
    fetch = 2.0 * fa.get(k, 0.0) / max(fc.get(k, 1), 1) * 1024  # gfx950: FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md)
    write = wa.get(k, 0.0) / max(wc.get(k, 1), 1) * 1024
    e = {"hbm_bytes_per_dispatch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write), "dispatches_fetch_pass": fc.get(k, 0)}
    if k in dur:
        ns, calls = dur[k]
        e.update({"mean_us_per_dispatch": round(ns / 1e3, 2), "calls_in_stats_run": calls,
                  "hbm_GBps": round((fetch + write) / ns, 1), "hbm_frac_of_8TBps": round((fetch + write) / ns / 8000.0, 4)})
    res[k] = e
json.dump({"_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) and --kernel-trace --stats on `python3 "
                          "scripts/fem_bench.py` (512 envs x 1920 tets: Dirichlet bench + the 30-step FemGelpad scene), MI355X; counter unit KiB, "
                          "FETCH_SIZE doubled on gfx950; means per dispatch; built by scripts/make_pmc_traffic_fem.py",
           "measured_at_commit": commit, "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
