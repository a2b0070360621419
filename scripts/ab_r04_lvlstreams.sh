#!/bin/bash
# band levels: two chunks in flight (TACEX_LEVEL_STREAMS=2, second stream) against one, at several chunk sizes; one box, two rounds
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04lvl}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "1 256" "2 128" "3 86" "4 64" "3 128" "4 128" "4 86"; do
    set -- $cfg
    TACEX_LEVEL_STREAMS=$1 TACEX_LEVEL_CHUNK_FRAMES=$2 timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $1 lcf $2 rep $rep: C3', d['value'], d['ms_per_step'], 'tail', d['roofline']['kernel_avg_ms'])" | tee -a $OUT/ab.log
  done
done
