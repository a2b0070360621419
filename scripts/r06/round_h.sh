#!/bin/bash
# fused FEM marker flow: tests, C4 / C5 sweep entries
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fem_gpu.py -x -q -k "marker" > gpurun_out/r06_h_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r06_h_tests.log
tail -15 gpurun_out/r06_h_tests.log
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --sweep-keys c4,c4_one_stream,c5,c4_ball --details-out gpurun_out/r06_h_details.json > gpurun_out/r06_h_bench.log 2>&1 || true
tail -1 gpurun_out/r06_h_bench.log | grep -o '"value_c4[^,]*\|"value_c5[^,]*'
