#!/bin/bash
# streaming tail with persistent waves (item counter) against the frozen library of the commit before, one box, alternating
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04persist}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "base 1" " 1" " 0"; do
    set -- $cfg; tag=$1; per=$2
    if [ -z "$per" ]; then per=$tag; tag=""; fi
    echo "== lib tag [$tag] TACEX_STREAM_PERSIST=$per rep $rep" | tee -a $OUT/ab.log
    TACEX_STREAM_PERSIST=$per TACEX_LIB_TAG=$tag TACEX_LIB_FROZEN=1 timeout 600 python scripts/tail_bench.py 1024 1 2>&1 | grep "^B=" | tee -a $OUT/ab.log
    TACEX_STREAM_PERSIST=$per TACEX_LIB_TAG=$tag TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline']['stages'].items()})" | tee -a $OUT/ab.log
  done
done
