#!/bin/bash
# streaming tail: RGB stores of a row issued one iteration later, behind the next row's fetches (late) against the product
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05late}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
TACEX_LIB_TAG=late TACEX_LIB_FROZEN=1 timeout 1200 python -m pytest tests/test_taxim_gpu.py tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py tests/test_sensor_configs_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/test.log
bash scripts/ab_r05.sh ${1:-r05late} "_ late" 3
for tag in _ late; do t=$tag; [ "$tag" = "_" ] && t=""
TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --sweep-keys c3_dense,c2,c5_optical,shard512 --details-out $OUT/d2.json > /dev/null 2>$OUT/err2.log; echo "[$tag]" | tee -a $OUT/sweep.log; python scripts/print_sweep.py $OUT/d2.json | cut -c1-60 | tee -a $OUT/sweep.log
done
