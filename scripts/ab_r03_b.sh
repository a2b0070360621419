#!/bin/bash
# round-3 A/B 2: occupancy of the base streaming tail (1 vs 2 workgroups per CU), clock split
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
run() { echo "== $1" | tee -a $O/out.txt; shift; env "$@" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt; }
run "base <9,5,3,5> 2 workgroups/CU" A=1
run "base, 1 workgroup per CU (LDS pad 24 KB)" TACEX_STREAM_LDS_PAD=24576
run "base, 1 workgroup per CU, 1 segment" TACEX_STREAM_LDS_PAD=24576 TACEX_STREAM_SEGS=1
run "base 2 workgroups/CU, 1 segment" TACEX_STREAM_SEGS=1
run "base 2 workgroups/CU, 4 segments" TACEX_STREAM_SEGS=4
