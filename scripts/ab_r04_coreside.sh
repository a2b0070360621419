#!/bin/bash
# two sensors on two streams with chunked passes: does a half-resident tail of one sensor leave room for the other's band levels?
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; env "$@" timeout 600 python bench.py $EXTRA --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  EXTRA="" run "one stream" X=1
  EXTRA="--sensor-streams" run "sensor streams" X=1
  EXTRA="--sensor-streams" run "sensor streams, passes of 256, 2 segments" TACEX_CHUNK_FRAMES=256 TACEX_STREAM_SEGS=2
  EXTRA="--sensor-streams" run "sensor streams, passes of 256, auto segments" TACEX_CHUNK_FRAMES=256
  EXTRA="--sensor-streams" run "sensor streams, passes of 512, 1 segment" TACEX_CHUNK_FRAMES=512 TACEX_STREAM_SEGS=1
  EXTRA="--sensor-streams" run "sensor streams, passes of 512, 2 segments" TACEX_CHUNK_FRAMES=512 TACEX_STREAM_SEGS=2
  EXTRA="--sensor-streams" run "sensor streams, passes of 128, 4 segments" TACEX_CHUNK_FRAMES=128 TACEX_STREAM_SEGS=4
  EXTRA="" run "one stream, passes of 256, 2 segments" TACEX_CHUNK_FRAMES=256 TACEX_STREAM_SEGS=2
done
