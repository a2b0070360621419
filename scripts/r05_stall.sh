#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05stall}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
echo "[default]" | tee -a $OUT/stall.log; timeout 600 python scripts/fem_stall_probe.py 150 2>&1 | grep -v "^W20\|^E20" | tail -4 | tee -a $OUT/stall.log
echo "[HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0]" | tee -a $OUT/stall.log; HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 timeout 600 python scripts/fem_stall_probe.py 150 2>&1 | grep -v "^W20\|^E20" | tail -4 | tee -a $OUT/stall.log
done
