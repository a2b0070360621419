#!/bin/bash
# SQ / TCP counters of the kernels matching <kernel-substring>: scripts/pmc_kernel.sh <tag> <kernel-substring> -- <abs script path> <args...>
TAG=$1; FILT=$2; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 "$@" > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $OUT/sq2 -- python3 "$@" > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT --output-format csv -d $OUT/sq3 -- python3 "$@" > $OUT/sq3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- python3 "$@" > $OUT/tcp.log 2>&1
cd $GRAFT_REPO_ROOT
for d in sq1 sq2 sq3 tcp; do python3 scripts/pmc_summary.py $OUT/$d "$FILT"; done
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
