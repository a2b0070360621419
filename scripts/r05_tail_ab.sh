#!/bin/bash
# streaming-tail variants on one box: bit-equality tests of the product library, then alternating timings
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05tail}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_taxim_gpu.py tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/test.log
bash scripts/ab_r05.sh ${1:-r05tail} "${2:-base _}" ${3:-2}
