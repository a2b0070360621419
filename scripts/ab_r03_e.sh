#!/bin/bash
# round-3 A/B 5: item order keyed by measured cost
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
run() { echo "== $1" | tee -a $O/out.txt; shift; env "$@" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-400 | tee -a $O/out.txt; }
run "frame order" TACEX_STREAM_ORDER=0
run "sorted by geometric key" TACEX_STREAM_ORDER_COST=0
run "sorted by measured cost" A=1
run "sorted by measured cost, 3 segments" TACEX_STREAM_SEGS=3
run "sorted by measured cost, 4 segments" TACEX_STREAM_SEGS=4
run "sorted by measured cost, 1 segment" TACEX_STREAM_SEGS=1
echo "== C3 bench" | tee -a $O/out.txt
python bench.py --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:round(v['avg_ms'],4) for k,v in d['roofline']['stages'].items()})" | tee -a $O/out.txt
TACEX_STREAM_ORDER=0 python bench.py --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('frame order', d['ms_per_step'], d['value'], {k:round(v['avg_ms'],4) for k,v in d['roofline']['stages'].items()})" | tee -a $O/out.txt
