#!/bin/bash
# small shards (C2: 256 frames, 512-env shard): band-level chunk sizes that put them on the two-streams + interleaved-depth path
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05small}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
run() { lab=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 10 --sweep-keys c2,c2_markers,shard512,c4_one_stream --sweep-steps 40 --details-out $OUT/d2.json > /dev/null 2>$OUT/err2.log
  echo "[$lab]" | tee -a $OUT/sweep.log; python scripts/print_sweep.py $OUT/d2.json | grep -v headline | cut -c1-60 | tee -a $OUT/sweep.log
}
for rep in 1 2; do
  run "default" A=1
  run "lcf=128" TACEX_LEVEL_CHUNK_FRAMES=128
  run "lcf=64 min=64" TACEX_LEVEL_CHUNK_FRAMES=64 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=64
  run "lcf=128 no-interleave" TACEX_LEVEL_CHUNK_FRAMES=128 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=4096
done
