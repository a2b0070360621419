#!/bin/bash
# round-6 GPU batch B: full GPU suite, ball scene with / without the coarse correction
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
for rep in 1 2; do for v in 1 0; do
  echo "TACEX_BALL_COARSE=$v" | tee -a $OUT/ball_coarse.log
  TACEX_BALL_COARSE=$v timeout 300 python scripts/r06/ball_probe.py 128 2>/dev/null | awk '{ms+=$3; nw+=$7; n++} END {printf "  30 steps: %.1f ms total, newton mean %.2f\n", ms, nw/n}' | tee -a $OUT/ball_coarse.log
  TACEX_BALL_COARSE=$v timeout 600 python bench.py --sweep-keys c4_ball --no-cpu-baseline --no-node-leg --steps 10 --details-out $OUT/d.json > /dev/null 2>$OUT/err.log; python scripts/print_sweep.py $OUT/d.json | grep c4_ball | cut -c1-160 | tee -a $OUT/ball_coarse.log
done; done
