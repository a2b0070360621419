#!/bin/bash
# ball kernel: every PCG vector in LDS + atomic coarse restriction (product) against the chains build (frozen tag `chains`)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fem_ball_gpu.py -x -q 2>&1 | tail -12 | tee $OUT/ball_test.log
for rep in 1 2 3; do for tag in _ chains; do t=$tag; [ "$tag" = "_" ] && t=""
  TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --sweep-keys c4_ball --no-cpu-baseline --no-node-leg --steps 10 --details-out $OUT/d.json > /dev/null 2>$OUT/err.log
  echo "[$tag] $(python scripts/print_sweep.py $OUT/d.json | grep c4_ball | cut -c1-170)" | tee -a $OUT/ab.log
done; done
