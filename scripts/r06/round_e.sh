#!/bin/bash
# edge-edge pairs: GPU parity tests of the ball scene, the probe with and without the pair kind, the bench key
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fem_ball_gpu.py -x -q > gpurun_out/r06_e_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r06_e_tests.log
tail -5 gpurun_out/r06_e_tests.log
PYTHONPATH=. timeout 300 python scripts/r06/ball_probe.py 512 > gpurun_out/r06_e_probe_ee.log 2>&1
tail -4 gpurun_out/r06_e_probe_ee.log
TACEX_BALL_EDGE_EDGE=0 PYTHONPATH=. timeout 300 python scripts/r06/ball_probe.py 512 > gpurun_out/r06_e_probe_noee.log 2>&1
tail -4 gpurun_out/r06_e_probe_noee.log
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --sweep-keys c4_ball > gpurun_out/r06_e_bench.log 2>&1 || true
tail -2 gpurun_out/r06_e_bench.log | cut -c1-600
