#!/bin/bash
# ball scene: trace of the late / PCG-heavy Newton iterations (library variant btrace)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
TACEX_LIB_TAG=btrace TACEX_LIB_FROZEN=1 PYTHONPATH=. timeout 600 python scripts/r06/ball_trace.py 512 > gpurun_out/r06_f_trace.log 2>&1
tail -3 gpurun_out/r06_f_trace.log
