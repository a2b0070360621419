#!/usr/bin/env python3
"""Which source lines the scratch reloads of one loop serve.  Listing from `hipcc -S -gline-tables-only`.
usage: isa_spill_map.py file.s <kernel-substring> <loop-rank (0 = longest)>"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
rank = int(sys.argv[3])
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and pat in l and ":" in l)
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
items, labels, loc = [], {}, None
for l in src[start:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+\d+\s+(\d+)\s+(\d+)", s)
    if m:
        loc = int(m.group(1))
        continue
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m:
        labels[m.group(1)] = len(items)
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    items.append((s.split(";")[0].strip(), loc))
loops = []
for i, (s, _) in enumerate(items):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", s)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t <= i:
            loops.append((t, i))
loops.sort(key=lambda x: x[0] - x[1])
t, e = loops[rank]
print("loop", (t, e), "len", e - t)


def regs_of(text):
    r = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        r.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        r.add(int(a))
    return r


for i in range(t, e + 1):
    x, lc = items[i]
    if not x.startswith("scratch_load"):
        continue
    ops = x.split(None, 1)[1]
    dst = regs_of(ops.split(",")[0])
    off = re.search(r"offset:(\d+)", x)
    use = None
    for j in range(i + 1, min(i + 600, len(items))):
        y = items[j][0]
        if " " not in y:
            continue
        o = y.split(None, 1)[1]
        parts = o.split(",", 1)
        srcs = regs_of(parts[1]) if len(parts) > 1 else set()
        if y.startswith(("ds_write", "global_store", "scratch_store", "v_cmp", "ds_bpermute", "global_load", "ds_read")):
            srcs = regs_of(o) if y.startswith(("ds_write", "global_store", "scratch_store", "v_cmp")) else srcs | regs_of(parts[1] if len(parts) > 1 else "")
        if dst & srcs:
            use = (j - i, y, items[j][1])
            break
    print(f"@{i - t:5d} line {lc}: {x:55s} -> +{use[0] if use else '?'} line {use[2] if use else '?'}: {use[1] if use else ''}")
