#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05misc}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for segs in 0 1 2 3; do
  TACEX_STREAM_SEGS=$segs timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 40 --details-out $OUT/d.json 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); det=json.load(open('$OUT/d.json'))
print('[segs $segs] C3', d['value'], d['ms_per_step'], {k: (round(v['avg_ms']*1e3,1), v['frames_per_launch']) for k, v in det['roofline']['stages'].items()})" | tee -a $OUT/segs.log
done
timeout 600 python bench.py --no-cpu-baseline --steps 10 --sweep-keys c4,c4_one_stream,c4_rolling,c5,c5_optical --details-out $OUT/d2.json > /dev/null 2>$OUT/err2.log; python scripts/print_sweep.py $OUT/d2.json | tee $OUT/sweep.log
