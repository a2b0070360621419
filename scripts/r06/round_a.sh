#!/bin/bash
# round-6 GPU batch A: ball tests, batch-size probe, FEM 256-thread A/B, tail knock-out (contact statistics) A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06a; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fem_ball_gpu.py -x -q 2>&1 | tail -30 > $OUT/ball_test.log
timeout 300 python scripts/r06/batch_probe.py > $OUT/batch_probe.log 2>&1
for rep in 1 2; do for v in 0 1; do
  TACEX_FEM_NT256=$v timeout 300 python scripts/r06/fem_nt256_ab.py 5,6,4 512 2>/dev/null | tee -a $OUT/nt256.log
  TACEX_FEM_NT256=$v timeout 300 python scripts/r06/fem_nt256_ab.py 5,6,4 256 2>/dev/null | tee -a $OUT/nt256.log
done; done
bash scripts/ab_r05.sh r06a "_ nostats" 3
tail -30 $OUT/ball_test.log; cat $OUT/batch_probe.log; cat $OUT/nt256.log; cat $OUT/ab.log
