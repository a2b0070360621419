#!/bin/bash
# depth -> height map pass: batched loads + DPP row reductions + no divisions (product) against the round's earlier kernel (u4)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05depth}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_taxim_gpu.py tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py tests/test_sensor_configs_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/test.log
for tag in u4 _ u4 _; do t=$tag; [ "$tag" = "_" ] && t=""
  echo "[$tag]" | tee -a $OUT/depth.log; TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 300 python scripts/depth_bench.py 2>&1 | tail -2 | tee -a $OUT/depth.log
done
bash scripts/ab_r05.sh ${1:-r05depth} "u4 _" 3
