#!/bin/bash
# 640x480: band-level chunk sizes with the depth pass interleaved (needs chunks of >= TACEX_DEPTH_INTERLEAVE_MIN_FRAMES)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05c5}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
run() { lab=$1; shift
  env "$@" timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 20 --height 480 --width 640 --envs-per-gpu 1024 --sensors 1 --details-out $OUT/d3.json 2>$OUT/err3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$lab] 640x480', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
}
for rep in 1 2; do
  run "default" A=1
  run "lcf=64 interleaved" TACEX_LEVEL_CHUNK_FRAMES=64 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=64
  run "lcf=64 upfront" TACEX_LEVEL_CHUNK_FRAMES=64
  run "lcf=128 interleaved" TACEX_LEVEL_CHUNK_FRAMES=128
  run "lcf=128 upfront" TACEX_LEVEL_CHUNK_FRAMES=128 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=4096
  run "lcf=32 interleaved" TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=32
done
