#!/bin/bash
# depth pass deferred into the render (default) against the two separate launches (TACEX_DEFER_DEPTH=0), same library; chunk knobs
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05defer}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_sensor_gpu.py tests/test_taxim_gpu.py tests/test_edge_cases_gpu.py tests/test_sensor_configs_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/test.log
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 --details-out $OUT/d.json 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); det=json.load(open('$OUT/d.json'))
print('[$lab] C3', d['value'], d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in det['roofline']['stages'].items()})" | tee -a $OUT/ab.log
}
for rep in 1 2; do
  run "defer=0" TACEX_DEFER_DEPTH=0
  run "defer=1" TACEX_DEFER_DEPTH=1
  run "defer=1 lcf=96" TACEX_LEVEL_CHUNK_FRAMES=96 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=64
  run "defer=1 lcf=192" TACEX_LEVEL_CHUNK_FRAMES=192
  run "defer=1 streams=3" TACEX_LEVEL_STREAMS=3 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=64
  run "defer=1 streams=4" TACEX_LEVEL_STREAMS=4 TACEX_DEPTH_INTERLEAVE_MIN_FRAMES=64
done
for rep in 1 2; do for d in 0 1; do
  TACEX_DEFER_DEPTH=$d timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 20 --height 480 --width 640 --envs-per-gpu 1024 --sensors 1 --details-out $OUT/d3.json 2>$OUT/err3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[defer=$d] rep $rep 640x480', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done; done
