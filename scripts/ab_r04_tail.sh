#!/bin/bash
# Taxim A/B on one box: frozen baseline library (TACEX_LIB_TAG=base TACEX_LIB_FROZEN=1) against the product, alternating
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04tail}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for tag in base mid ""; do
    echo "== lib tag [$tag] rep $rep" | tee -a $OUT/ab.log
    TACEX_LIB_TAG=$tag TACEX_LIB_FROZEN=1 timeout 600 python scripts/tail_bench.py 1024 1 2>&1 | grep "^B=" | tee -a $OUT/ab.log
    TACEX_LIB_TAG=$tag TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline']['stages'].items()})" | tee -a $OUT/ab.log
  done
done
