#!/bin/bash
# ball scene: trace of the straggler envs' late Newton iterations; the scene at 4096 envs (amortised stragglers)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
TACEX_LIB_TAG=btrace TACEX_LIB_FROZEN=1 PYTHONPATH=. timeout 600 python scripts/r06/ball_trace.py 512 > gpurun_out/r06_f_trace.log 2>&1
tail -3 gpurun_out/r06_f_trace.log
for B in 1024 2048 4096; do
  PYTHONPATH=. timeout 600 python scripts/r06/ball_probe.py $B > gpurun_out/r06_f_probe_$B.log 2>&1
  tail -2 gpurun_out/r06_f_probe_$B.log
done
