#!/bin/bash
# A/B of the headline bench (C3, no sweep / cpu baseline) across library build flags; each variant rebuilds on the GPU box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
for v in "$@"; do
  for rep in 1 2; do
    echo -n "[$v] rep $rep: " | tee -a gpurun_out/variants/bench.txt
    TACEX_EXTRA_HIPCC_FLAGS="$v" python bench.py --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:round(v['avg_ms'],4) for k,v in d['roofline']['stages'].items()})" | tee -a gpurun_out/variants/bench.txt
  done
done
