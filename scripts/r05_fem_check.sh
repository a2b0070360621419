#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05fem}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fem_physics_gpu.py -x -q -m gpu -s 2>&1 | tail -40 | tee $OUT/physics.log
timeout 1500 python -m pytest tests/test_fem_gpu.py -q -m gpu 2>&1 | tail -30 | tee $OUT/fem.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --details-out $OUT/d.json > $OUT/line.json 2> $OUT/err.log
python scripts/print_sweep.py $OUT/d.json | tee $OUT/sweep.log
