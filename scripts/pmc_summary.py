"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch.
   Usage: python scripts/pmc_summary.py <dir-or-csv> [name-filter]"""
import csv, glob, os, sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if flt and flt not in k:
                continue
            k = k[:70]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:32s} {acc[k][c] / cnt[k][c]:16.1f}  (n={cnt[k][c]})")
