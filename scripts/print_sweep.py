#!/usr/bin/env python3
"""One line per sweep entry of a bench details file (bench.py --details-out)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print("headline", d["value"], d["ms_per_step"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d.get("roofline", {}).get("stages", {}).items()})
for e in d["config"].get("sweep", []):
    per = e.get("fem_period") or {}
    print(e["key"], e.get("frames_per_s", e.get("env_steps_per_s")), e.get("ms_per_step"), "fem_ms", e.get("fem_ms_mean"), e.get("fem_ms_min_max"),
          "newton/step", per.get("newton_iters_per_step_mean"), "pcg/newton", per.get("pcg_iters_per_newton_mean"), "max_iters", e.get("newton_iters_max_over_period", e.get("newton_iters_max")),
          e.get("error", ""))
