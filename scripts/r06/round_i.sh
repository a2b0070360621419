#!/bin/bash
# blocks assembled ahead + fused marker flow: FEM tests, C4 / C5 entries, C4 timeline
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fem_gpu.py tests/test_fem_physics_gpu.py -x -q > gpurun_out/r06_i_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r06_i_tests.log
tail -6 gpurun_out/r06_i_tests.log
timeout 900 python bench.py --no-node-leg --no-cpu-baseline --sweep-keys c4,c4_one_stream,c4_rolling,c5 --details-out gpurun_out/r06_i_details.json > gpurun_out/r06_i_bench.log 2>&1 || true
tail -1 gpurun_out/r06_i_bench.log | grep -o '"value_c4[^,]*\|"value_c5[^,]*'
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c4 -- python3 scripts/r06/c4_timeline.py > gpurun_out/r06_c4_timeline_run.log 2>&1
grep "C4:" gpurun_out/r06_c4_timeline_run.log
python3 scripts/r06/timeline_of.py gpurun_out/prof_c4 fem_newton_lds_kernel 4 > gpurun_out/r06_c4_step_timeline.txt
rm -rf gpurun_out/prof_c4
