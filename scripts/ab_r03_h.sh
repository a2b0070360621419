#!/bin/bash
# round-3 probe: memory-system latency / stall counters of the streaming tail (tail_bench at 1024 frames)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pm() { local name=$1; local ctr=$2; shift 2
  env "$@" timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -- python3 $GRAFT_REPO_ROOT/scripts/tail_bench.py 1024 1 > $O/$name.log 2>&1; echo "$name rc=$?"; }
pm l1lat "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" A=1
pm ealat "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" A=1
pm eawlat "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" A=1
pm l2hit "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum" A=1
pm l2stall "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_LATENCY_FIFO_FULL_sum" A=1
pm credit "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_IB_STALL_sum" A=1
pm tcp "TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" A=1
pm vmemlat "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACCUM_PREV_HIRES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" A=1
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03h"
for d in sorted(glob.glob(O+"/*/")):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "taxim_stream" in k or "frame_rows" in k or "blur_mfma_kernel<33" in k:
                acc[k[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(os.path.basename(d.rstrip("/")), k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
