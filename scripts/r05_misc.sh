#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05misc}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for segs in 0 2 3 4; do
  TACEX_STREAM_SEGS=$segs timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 --details-out $OUT/d.json 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); det=json.load(open('$OUT/d.json'))
print('[segs $segs] C3', d['value'], d['ms_per_step'], {k: (round(v['avg_ms']*1e3,1), v['frames_per_launch']) for k, v in det['roofline']['stages'].items()})" | tee -a $OUT/segs.log
done
