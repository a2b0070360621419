#!/bin/bash
# round-6 GPU batch C: ball tests, c4_ball at 512 envs with one / two envs co-resident per CU (frozen variant wg2)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fem_ball_gpu.py -x -q 2>&1 | tail -12 | tee $OUT/ball_test.log
for rep in 1 2; do for tag in _ wg2; do t=$tag; [ "$tag" = "_" ] && t=""
  TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --sweep-keys c4_ball --no-cpu-baseline --no-node-leg --steps 10 --details-out $OUT/d.json > /dev/null 2>$OUT/err.log
  echo "[$tag] $(python scripts/print_sweep.py $OUT/d.json | grep c4_ball | cut -c1-170)" | tee -a $OUT/wg.log
done; done
