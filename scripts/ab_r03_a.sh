#!/bin/bash
# round-3 A/B 1: streaming tail occupancy and the three-level tail (prebuilt tagged libraries, see tacex_amd/_build.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
run() { echo "== $1" | tee -a $O/out.txt; shift; env "$@" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt; }
run "base <9,5,3,5> 2 waves/SIMD" A=1
run "base, 1 workgroup per CU (LDS pad)" TACEX_STREAM_LDS_PAD=8192
run "<5,3,5> 3 waves/SIMD (scratch spills) + k9 band" TACEX_TAIL_LEVELS_320=3
run "<5,3,5> 3 waves, 2 segments" TACEX_TAIL_LEVELS_320=3 TACEX_STREAM_SEGS=2
run "<5,3,5> 2 waves/SIMD + k9 band" TACEX_TAIL_LEVELS_320=3 TACEX_LIB_TAG=w2
run "base again" A=1
echo "== parity of the 3-level variant" | tee -a $O/out.txt
TACEX_TAIL_LEVELS_320=3 timeout 900 python -m pytest tests/test_taxim_gpu.py tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee -a $O/out.txt
