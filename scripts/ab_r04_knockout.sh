#!/bin/bash
# Which vector-memory operation the shading waits for: the streaming tail with one class of memory instruction compiled out at a time
# (TACEX_DBG_NO_STORE / NO_BG / NO_GATHER / NO_ROWLOAD; wrong images, same arithmetic), fused and as the split levels / shading pair.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04ko}; mkdir -p $OUT
for tag in "" dbg_NO_STORE dbg_NO_BG dbg_NO_GATHER dbg_NO_ROWLOAD dbg_ALL; do
  echo "== [$tag] fused" | tee -a $OUT/ko.log
  (cd $GRAFT_REPO_ROOT && TACEX_LIB_TAG=$tag python scripts/tail_bench.py 1024 1 2>&1 | grep "^B=" | sed 's/.*tail_fused=/tail_fused=/' | tee -a $OUT/ko.log)
  (cd /tmp && export TMPDIR=/tmp && TACEX_LIB_TAG=$tag TACEX_STREAM_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$tag -- python3 $GRAFT_REPO_ROOT/scripts/tail_bench.py 1024 1 > $OUT/p_$tag.log 2>&1)
  echo "== [$tag] split pair" | tee -a $OUT/ko.log
  grep "taxim_stream_kernel" $(find $OUT/p_$tag -name "*kernel_stats.csv") | awk -F, '{print $1, $4}' | cut -c1-120 | tee -a $OUT/ko.log
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
