#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05misc}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for vt in 1e-7; do
  TACEX_TEST_VTOL=$vt TACEX_TEST_REPORT=1 timeout 900 python -m pytest tests/test_fem_physics_gpu.py -x -q -m gpu -s -k "stationary and 0.001" 2>&1 | grep "^step\|^d_hat\|passed\|failed\|(c)" > $OUT/physics_report_$vt.log; tail -3 $OUT/physics_report_$vt.log
done
