#!/bin/bash
# band-level streams at a 512-env shard (one sensor) and under the FEM side stream of C4
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for s in 1 2; do
    TACEX_LEVEL_STREAMS=$s timeout 600 python bench.py --envs-per-gpu 512 --sensors 1 --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512 x 1 sensor, level streams $s:', d['value'], d['ms_per_step'])"
    TACEX_LEVEL_STREAMS=$s timeout 600 python bench.py --envs-per-gpu 256 --sensors 1 --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256 x 1 sensor, level streams $s:', d['value'], d['ms_per_step'])"
    echo "level streams $s:"; TACEX_LEVEL_STREAMS=$s timeout 600 python scripts/c4_quick.py 2>&1 | grep -v amdgpu | head -4
  done
done
