#!/bin/bash
# memory-system counters of the streaming tail (small counter sets per pass, --kernel-trace only): what its 12 B/px of RGB stores run into
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05tcc}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL" "TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL TCC_BUSY" "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TA_BUSY GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-roofline > $OUT/p$i.log 2>&1; echo "p$i rc=$? ($set)"
done
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*_agent_info.csv" -delete 2>/dev/null; du -sh $OUT
