#!/bin/bash
# round-3 A/B 6: compacted table gather vs per-slot gather, with / without the sorted item order
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
run() { echo "== $1" | tee -a $O/out.txt; shift; env "$@" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-400 | tee -a $O/out.txt; }
run "per-slot gather, geometric order" TACEX_LIB_TAG=nc TACEX_STREAM_ORDER_COST=0
run "compacted gather, geometric order" TACEX_STREAM_ORDER_COST=0
run "compacted gather, frame order" TACEX_STREAM_ORDER=0
run "compacted gather, measured-cost order" A=1
run "per-slot gather, geometric order (again)" TACEX_LIB_TAG=nc TACEX_STREAM_ORDER_COST=0
run "compacted gather, geometric order (again)" TACEX_STREAM_ORDER_COST=0
echo "== gpu tests (taxim, sensor, edge cases)" | tee -a $O/out.txt
timeout 1500 python -m pytest tests/test_taxim_gpu.py tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee -a $O/out.txt
