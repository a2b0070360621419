#!/bin/bash
# round-3 A/B 4: item order of the streaming tail (heaviest first) vs frame order, segment counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
run() { echo "== $1" | tee -a $O/out.txt; shift; env "$@" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt; }
run "frame order, 2 segments" TACEX_STREAM_ORDER=0
run "sorted, 2 segments" A=1
run "sorted, 3 segments" TACEX_STREAM_SEGS=3
run "sorted, 4 segments" TACEX_STREAM_SEGS=4
run "frame order, 4 segments" TACEX_STREAM_ORDER=0 TACEX_STREAM_SEGS=4
run "sorted, 1 segment" TACEX_STREAM_SEGS=1
echo "== gpu tests" | tee -a $O/out.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee -a $O/out.txt
