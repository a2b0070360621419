#!/bin/bash
# Newton kernel: E(x) of the line search from the gradient sweep's tet states (product) against a tet sweep of its own (sepe0)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05e0}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fem_gpu.py tests/test_fem_physics_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/test.log
for rep in 1 2; do for tag in sepe0 _; do t=$tag; [ "$tag" = "_" ] && t=""
echo "[$tag] rep $rep" | tee -a $OUT/ab.log
TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 300 python scripts/fem_bench.py 2>&1 | grep "FemGelpad scene" | tee -a $OUT/ab.log
TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --sweep-keys c4,c4_one_stream,c4_rolling,c5,axle --details-out $OUT/d2.json > /dev/null 2>$OUT/err2.log; python scripts/print_sweep.py $OUT/d2.json | grep -v headline | cut -c1-150 | tee -a $OUT/ab.log
done; done
