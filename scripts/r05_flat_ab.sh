#!/bin/bash
# frames without contact on the flat path of the streaming tail (product) against marching them (prev)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05flat}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_taxim_gpu.py tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py tests/test_sensor_configs_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/test.log
bash scripts/ab_r05.sh ${1:-r05flat} "prev _ fb8 fb2" 3
for tag in prev _; do t=$tag; [ "$tag" = "_" ] && t=""
TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --sweep-keys c3_dense,c2,c2_markers,shard512,c5_optical,ref_scene --details-out $OUT/d2.json > /dev/null 2>$OUT/err2.log; echo "[$tag]" | tee -a $OUT/sweep.log; python scripts/print_sweep.py $OUT/d2.json | grep -v headline | cut -c1-50 | tee -a $OUT/sweep.log
done
