"""One steady step out of a rocprofv3 kernel trace: the kernels between the starts of two consecutive launches of the anchor kernel (argv[2], a
substring of its name), taken `argv[3]` anchors before the last; start / duration in us, hardware queue."""
import csv, glob, sys
root, anchor = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 4
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[-1]  # (prof keeps its trace: the "timeline" step of gpu_round.sh runs after "prof")
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[3]]
a, b = idx[-back - 1], idx[-back]
t0 = rows[a][0]
qs = {}
print(f"# {f}: kernels from launch {len(idx) - back - 1} to launch {len(idx) - back} of `{anchor}` ({(rows[b][0] - t0) / 1e3:.1f} us)")
print("    start      dur  q  kernel")
for s, e, q, name in rows[a:b]:
    qn = qs.setdefault(q, len(qs) + 1)
    short = name.split("(")[0].replace("void ", "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {qn}  {short[:110]}")
