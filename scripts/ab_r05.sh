#!/bin/bash
# Taxim A/B on one box: frozen library variants (TACEX_LIB_TAG=<tag> TACEX_LIB_FROZEN=1) against the product (""), alternating.
# usage: ab_r05.sh <outdir-name> "<tag list>" [reps]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05ab}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
TAGS=${2:-"base _"}
for rep in $(seq 1 ${3:-2}); do
  for tag in $TAGS; do
    t=$tag; [ "$tag" = "_" ] && t=""
    TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-sweep --no-node-leg --no-cpu-baseline --steps 40 --details-out $OUT/d.json 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); det=json.load(open('$OUT/d.json'))
print('[$tag] rep $rep C3', d['value'], d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in det['roofline']['stages'].items()})" | tee -a $OUT/ab.log
  done
done
