#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
for t in clk8 c8ns c8nb c8ng; do
  echo "== $t, 1 wave/SIMD" | tee -a $O/out.txt
  TACEX_LIB_TAG=$t TACEX_STREAM_LDS_PAD=24576 python scripts/stream_clock8.py 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
  echo "== $t, 2 waves/SIMD" | tee -a $O/out.txt
  TACEX_LIB_TAG=$t python scripts/stream_clock8.py 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
done
