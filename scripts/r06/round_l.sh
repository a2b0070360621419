#!/bin/bash
# ball kernel with the ball-side candidate lists: parity tests, phase clock, bench keys
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fem_ball_gpu.py -x -q > gpurun_out/r06_l_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r06_l_tests.log
tail -4 gpurun_out/r06_l_tests.log
TACEX_LIB_TAG=bclk TACEX_LIB_FROZEN=1 PYTHONPATH=. timeout 300 python scripts/r06/ball_clock.py 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_l_clock.log
tail -4 gpurun_out/r06_l_clock.log
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --no-roofline --sweep-keys c4_ball,c4_ball4096 --details-out gpurun_out/r06_l_details.json > gpurun_out/r06_l_bench.log 2>&1 || true
tail -1 gpurun_out/r06_l_bench.log | grep -o '"value_c4_ball[^,]*,"value_c4_ball4096[^,]*'
