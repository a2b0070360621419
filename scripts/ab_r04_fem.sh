#!/bin/bash
# FEM A/B on one box: the frozen baseline library (built from an older commit, TACEX_LIB_TAG=base TACEX_LIB_FROZEN=1) against the product
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04fem}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for tag in base ""; do
    echo "== lib tag [$tag] rep $rep" | tee -a $OUT/ab.log
    TACEX_LIB_TAG=$tag TACEX_LIB_FROZEN=1 timeout 600 python scripts/fem_bench.py 2>&1 | grep -v "^element_terms\|^energy\|^gradient" | tee -a $OUT/ab.log
  done
done
