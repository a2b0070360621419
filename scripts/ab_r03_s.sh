#!/bin/bash
# A/B: LLVM scheduling strategies for taxim_stream.hip (TACEX_STREAM_HIPCC_FLAGS, tagged libraries), tail at 1024 frames, two rounds
cd $GRAFT_REPO_ROOT
for r in 1 2; do for t in "" tbase tilp tmmc2 tmmc3; do
  if [ -n "$t" ] && [ ! -f tacex_amd/libtacex_hip.$t.so ]; then continue; fi
  echo "== tag '$t'"; TACEX_LIB_TAG=$t python scripts/tail_bench.py 1024 1 2>&1 | grep "obs=True" | sed 's/.*total/total/' | cut -c1-60,190-260
done; done
