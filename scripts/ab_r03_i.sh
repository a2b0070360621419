#!/bin/bash
# round-3 A/B 7: linear RGB stores (LDS transposition) vs the strided stores of round 2; GPU test suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
run() { echo "== $1" | tee -a $O/out.txt; shift; env "$@" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-400 | tee -a $O/out.txt; }
run "strided stores" TACEX_LIB_TAG=ss
run "linear stores" A=1
run "strided stores (again)" TACEX_LIB_TAG=ss
run "linear stores (again)" A=1
run "linear stores, frame order" TACEX_STREAM_ORDER=0
run "linear stores, measured-cost order off" TACEX_STREAM_ORDER_COST=0
echo "== C3 bench" | tee -a $O/out.txt
python bench.py --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:round(v['avg_ms'],4) for k,v in d['roofline']['stages'].items()})" | tee -a $O/out.txt
echo "== gpu tests" | tee -a $O/out.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee -a $O/out.txt
