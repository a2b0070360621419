#!/bin/bash
# item order of the tail issued beside the band levels (TACEX_STREAM_ORDER_EARLY=1, default) against behind their join (0)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05order}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sensor_gpu.py tests/test_edge_cases_gpu.py tests/test_taxim_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/test.log
for rep in 1 2 3; do for v in 0 1; do
  TACEX_STREAM_ORDER_EARLY=$v timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 --details-out $OUT/d.json 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[early=$v] rep $rep C3', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done; done
for v in 0 1; do
TACEX_STREAM_ORDER_EARLY=$v timeout 600 python bench.py --no-cpu-baseline --steps 10 --sweep-keys c3_separate,c5_optical --details-out $OUT/d2.json > /dev/null 2>$OUT/err2.log; echo "[early=$v]" | tee -a $OUT/ab.log; python scripts/print_sweep.py $OUT/d2.json | grep -v headline | cut -c1-50 | tee -a $OUT/ab.log
done
