#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05misc}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== default"; python scripts/fem_idle_probe.py 16 2>/dev/null | tee -a $OUT/idle_default.log
echo "== HSA_SCRATCH_SINGLE_LIMIT=1GB"; HSA_SCRATCH_SINGLE_LIMIT=1073741824 python scripts/fem_idle_probe.py 16 2>/dev/null | tee -a $OUT/idle_limit.log
done
