#!/bin/bash
# group launch A/B on one box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05group; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sensor_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/test.log
run() { # label, env, args
  for rep in 1 2; do
    env $2 timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 --details-out $OUT/d.json $3 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); det=json.load(open('$OUT/d.json'))
print('[$1] rep $rep C3', d['value'], d['ms_per_step'], {k: (round(v['avg_ms']*1e3,1), v['frames_per_launch'], v['launches_per_update']) for k, v in det['roofline']['stages'].items()})" | tee -a $OUT/ab.log
  done
}
run separate "A=1" "--no-group"
run group_1024 "A=1" ""
run group_2048 "TACEX_CHUNK_FRAMES=0" ""
run separate "A=1" "--no-group"
run group_2048 "TACEX_CHUNK_FRAMES=0" ""
