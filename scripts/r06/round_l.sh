#!/bin/bash
# ball kernel: parity tests, (phase clock when libtacex_hip.bclk.so was built with -DTACEX_BALL_CLOCK), probe, bench keys
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fem_ball_gpu.py -x -q > gpurun_out/r06_l_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r06_l_tests.log
tail -4 gpurun_out/r06_l_tests.log
if [ -f tacex_amd/libtacex_hip.bclk.so ]; then
  TACEX_LIB_TAG=bclk TACEX_LIB_FROZEN=1 PYTHONPATH=. timeout 300 python scripts/r06/ball_clock.py 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_l_clock.log
  tail -4 gpurun_out/r06_l_clock.log
fi
PYTHONPATH=. timeout 300 python scripts/r06/ball_probe.py 512 2>&1 | grep "^step" > gpurun_out/r06_l_probe.log
python3 - <<'PY'
import numpy as np
L=[l for l in open('gpurun_out/r06_l_probe.log')]
ms=[float(l.split()[2]) for l in L]; nw=[float(l.split('newton mean')[1].split()[0]) for l in L]; mx=[int(l.split('max')[1].split()[0]) for l in L]; pc=[float(l.split('pcg/newton')[1].split()[0]) for l in L]
print(f"probe steps 6-29: ms mean {np.mean(ms[6:]):.2f} (max {np.max(ms[6:]):.2f}); newton mean {np.mean(nw[6:]):.2f}, worst env {max(mx[6:])}; pcg/newton {np.mean(pc[6:]):.1f}")
PY
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --no-roofline --sweep-keys c4_ball,c4_ball4096 --details-out gpurun_out/r06_l_details.json > gpurun_out/r06_l_bench.log 2>&1 || true
tail -1 gpurun_out/r06_l_bench.log | grep -o '"value_c4_ball[^,]*,"value_c4_ball4096[^,]*'
