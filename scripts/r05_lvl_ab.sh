#!/bin/bash
# block-skip threshold per level (k34 / k61 / k117 = levels below that size compiled without the block tests) and the depth pass's unroll
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05lvl}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for tag in _ u1 u4 u5 _; do t=$tag; [ "$tag" = "_" ] && t=""
  echo "[$tag]" | tee -a $OUT/depth.log; TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 300 python scripts/depth_bench.py 2>&1 | tail -2 | tee -a $OUT/depth.log
done
bash scripts/ab_r05.sh ${1:-r05lvl} "_ k34 k61 k117 u4" 2
for rep in 1 2; do for tag in _ k34 k61 k117; do t=$tag; [ "$tag" = "_" ] && t=""
TACEX_LIB_TAG=$t TACEX_LIB_FROZEN=1 timeout 600 python bench.py --no-sweep --no-cpu-baseline --steps 20 --height 480 --width 640 --envs-per-gpu 1024 --sensors 1 --details-out $OUT/d3.json 2>$OUT/err3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); det=json.load(open('$OUT/d3.json'))
print('[$tag] rep $rep 640x480', d['value'], d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in det['roofline']['stages'].items()})" | tee -a $OUT/ab640.log
done; done
