#!/bin/bash
# ball kernel on 256 threads (one wave per SIMD, no scratch) against 512: tests, probe, bench keys; alternating
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
for nt in 256 512 256 512; do
  export TACEX_BALL_NT=$nt
  if [ $nt = 256 ] && [ ! -f gpurun_out/r06_n_tests_done ]; then
    timeout 1500 python -m pytest tests/test_fem_ball_gpu.py -x -q > gpurun_out/r06_n_tests_256.log 2>&1; echo "NT=256 tests exit $?"; tail -2 gpurun_out/r06_n_tests_256.log; touch gpurun_out/r06_n_tests_done
  fi
  PYTHONPATH=. timeout 300 python scripts/r06/ball_probe.py 512 2>&1 | grep "^step" > gpurun_out/r06_n_probe.log
  python3 - <<'PY'
import numpy as np, os
L=[l for l in open('gpurun_out/r06_n_probe.log')]
ms=[float(l.split()[2]) for l in L]; nw=[float(l.split('newton mean')[1].split()[0]) for l in L]; mx=[int(l.split('max')[1].split()[0]) for l in L]; pc=[float(l.split('pcg/newton')[1].split()[0]) for l in L]
print(f"NT={os.environ['TACEX_BALL_NT']} probe steps 6-29: ms mean {np.mean(ms[6:]):.2f} (max {np.max(ms[6:]):.2f}); newton mean {np.mean(nw[6:]):.2f}, worst env {max(mx[6:])}; pcg/newton {np.mean(pc[6:]):.1f}")
PY
  timeout 900 python bench.py --no-node-leg --no-cpu-baseline --no-roofline --sweep-keys c4_ball,c4_ball4096 --details-out gpurun_out/r06_n_details.json > gpurun_out/r06_n_bench.log 2>&1 || true
  echo "NT=$nt $(tail -1 gpurun_out/r06_n_bench.log | grep -o '"value_c4_ball[^,]*,"value_c4_ball4096[^,]*')"
done
rm -f gpurun_out/r06_n_tests_done
