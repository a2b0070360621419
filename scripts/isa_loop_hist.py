#!/usr/bin/env python3
"""Instruction histogram of the hot loop of one kernel in a hipcc -S listing.
usage: isa_loop_hist.py file.s <kernel-name-substring> [--dump]
The hot loop = the backward branch spanning the most instructions that contains a global_load_lds (or, failing that, the longest one)."""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and pat in l and l.rstrip().endswith(("E:", "_:")) or (l.startswith("_Z") and pat in l and ":" in l and "@" in l))
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i] or src[i].startswith("\t.section"))
body = src[start:end]
labels = {}
ins = []
for l in body:
    s = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    ins.append(s.split(";")[0].strip())
loops = []
for i, s in enumerate(ins):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", s)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t <= i:
            loops.append((t, i))
best = None
for t, i in loops:
    has = any("global_load_lds" in x or "s_waitcnt vmcnt" in x for x in ins[t:i + 1])
    key = (has, i - t)
    if best is None or key > best[0]:
        best = (key, t, i)
_, t, i = best
loop = ins[t:i + 1]
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"
c = Counter(); ops = Counter()
for s in loop:
    op = s.split()[0]
    c[cls(op)] += 1
    ops[op] += 1
print(f"kernel {pat}: {len(ins)} instructions, hot loop [{t},{i}] = {len(loop)} instructions; all loops: {[(a, b - a) for a, b in loops]}")
print(dict(c))
for op, n in ops.most_common(60):
    print(f"  {n:5d} {op}")
if "--dump" in sys.argv:
    print("\n".join(loop))
