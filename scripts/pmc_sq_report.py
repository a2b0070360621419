"""Markdown summary of a rocprofv3 --pmc SQ_* pass (scripts/gpu_round.sh pmcsq): per tacex kernel, mean counters per dispatch and
the ratios VERDICT asks for (LDS bank-conflict ratio, wait fractions).
   Usage: python scripts/pmc_sq_report.py <dir> <out.md> [commit] [command]"""
import csv, glob, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else "?"
cmd = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-roofline"
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "tacex::" not in k:
                continue
            k = k.replace("void ", "").split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
lines = [f"# SQ counters per dispatch (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
         f"SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS)", "",
         f"Command: `{cmd}` at commit `{commit}` (C3: 1024 envs x 2 sensors, 1024 frames per launch). Means over the dispatches of the run.", "",
         "| kernel | n | LDS conflict / LDS active | WAIT_ANY / WAVE_CYCLES | WAIT_INST_LDS / BUSY_CYCLES | VALU active / BUSY_CYCLES | WAVE_CYCLES | BUSY_CYCLES |",
         "|---|---|---|---|---|---|---|---|"]
for k in sorted(acc):
    m = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    n = max(cnt[k].values())
    r = lambda a, b: f"{m.get(a, 0) / m[b]:.3f}" if m.get(b) else "-"
    lines.append(f"| `{k}` | {n} | {r('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')} | {r('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')} | "
                 f"{r('SQ_WAIT_INST_LDS', 'SQ_BUSY_CYCLES')} | {r('SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES')} | {m.get('SQ_WAVE_CYCLES', 0):.3g} | {m.get('SQ_BUSY_CYCLES', 0):.3g} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
