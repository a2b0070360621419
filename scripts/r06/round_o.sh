#!/bin/bash
# line-search refinement (cfg.line_search.refine = bisections after a cut step): ball tests at the default, then the probe at 0 / 2 / 4 / 6 alternating
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fem_ball_gpu.py -x -q > gpurun_out/r06_o_tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/r06_o_tests.log
for rf in 0 4 0 2 6; do
  export TACEX_BALL_LS_REFINE=$rf
  PYTHONPATH=. timeout 300 python scripts/r06/ball_probe.py 512 2>&1 | grep "^step" > gpurun_out/r06_o_probe.log
  python3 - <<'PY'
import numpy as np, os
L=[l for l in open('gpurun_out/r06_o_probe.log')]
ms=[float(l.split()[2]) for l in L]; nw=[float(l.split('newton mean')[1].split()[0]) for l in L]; mx=[int(l.split('max')[1].split()[0]) for l in L]; pc=[float(l.split('pcg/newton')[1].split()[0]) for l in L]; fl=[int(l.split('flags')[1].split()[0]) for l in L]
print(f"refine={os.environ['TACEX_BALL_LS_REFINE']} probe steps 6-29: ms mean {np.mean(ms[6:]):.2f} (max {np.max(ms[6:]):.2f}); newton mean {np.mean(nw[6:]):.2f}, worst env {max(mx[6:])} (mean of per-step worst {np.mean(mx[6:]):.1f}); pcg/newton {np.mean(pc[6:]):.1f}; flags {max(fl)}")
PY
done
unset TACEX_BALL_LS_REFINE
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --no-roofline --sweep-keys c4_ball,c4_ball4096 > gpurun_out/r06_o_bench.log 2>&1 || true
echo "default cfg: $(tail -1 gpurun_out/r06_o_bench.log | grep -o '"value_c4_ball[^,]*,"value_c4_ball4096[^,]*')"
