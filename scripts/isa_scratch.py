#!/usr/bin/env python3
"""Where a kernel touches its spill scratch: per loop of a hipcc -S listing, the scratch loads / stores, barriers and f64 ops,
and (with --sites LOOPINDEX) the scratch / global loads and waits of that loop in program order.
usage: isa_scratch.py file.s <kernel-name-substring> [--sites N]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and pat in l and ":" in l and not l.startswith("\t"))
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i] or src[i].startswith("\t.section"))
ins, labels = [], {}
for l in src[start:end]:
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
loops.sort(key=lambda x: x[0] - x[1])
print(len(ins), "instructions; scratch loads", sum(x.startswith("scratch_load") for x in ins), "stores", sum(x.startswith("scratch_store") for x in ins))
for n, (t, i) in enumerate(loops[:12]):
    seg = ins[t:i + 1]
    print(n, (t, i), "len", i - t, "scratch ld/st", sum(x.startswith("scratch_load") for x in seg), sum(x.startswith("scratch_store") for x in seg),
          "barriers", sum(x.startswith("s_barrier") for x in seg), "f64", sum(bool(re.match(r"v_(fma|mul|add)_f64", x)) for x in seg),
          "readlane", sum("v_readlane" in x for x in seg), "accvgpr", sum("accvgpr" in x for x in seg))
if "--sites" in sys.argv:
    t, i = loops[int(sys.argv[sys.argv.index("--sites") + 1])]
    for k, x in enumerate(ins[t:i + 1]):
        if x.startswith(("scratch_", "s_barrier", "global_load", "global_store")) or "s_waitcnt vmcnt" in x:
            print(f"{k}: {x}")
