#!/bin/bash
# round-3 probe 3: dynamic instruction mix / issue-stall split of the streaming tail (SQ counters, tail_bench at 1024 frames)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pm() { # name, counters, env...
  local name=$1; local ctr=$2; shift 2
  env "$@" timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -- python3 $GRAFT_REPO_ROOT/scripts/tail_bench.py 1024 1 > $O/$name.log 2>&1
  echo "$name rc=$?"
}
pm base_insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES" A=1
pm base_cyc "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" A=1
pm base_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" A=1
pm l3_insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES" TACEX_TAIL_LEVELS_320=3 TACEX_LIB_TAG=w2
pm l3_cyc "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" TACEX_TAIL_LEVELS_320=3 TACEX_LIB_TAG=w2
pm base_ta "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum" A=1
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03c"
for d in sorted(glob.glob(O+"/*/")):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "taxim_stream" in k or "blur_mfma_kernel<9" in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(os.path.basename(d.rstrip("/")), k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
