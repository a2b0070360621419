#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fem_ball_gpu.py -x -q > gpurun_out/r06_g_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r06_g_tests.log
tail -4 gpurun_out/r06_g_tests.log
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --sweep-keys c4_ball,c4_ball4096 --details-out gpurun_out/r06_g_details.json > gpurun_out/r06_g_bench.log 2>&1 || true
tail -1 gpurun_out/r06_g_bench.log | grep -o '"value_c4_ball[^,]*,"value_c4_ball4096[^,]*'
