#!/bin/bash
# A/B timing of the streaming tail with debug macros (each variant rebuilds the library on the GPU box).
# usage: scripts/stream_variants.sh ["flagsA" "flagsB" ...]   (default: the memory-op removal set)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
if [ $# -eq 0 ]; then
  set -- "" "-DTACEX_DBG_NO_GATHER" "-DTACEX_DBG_NO_STORE" "-DTACEX_DBG_NO_BG" "-DTACEX_DBG_NO_ROWLOAD" "-DTACEX_DBG_NO_GATHER -DTACEX_DBG_NO_STORE -DTACEX_DBG_NO_BG -DTACEX_DBG_NO_ROWLOAD"
fi
for v in "$@"; do
  echo "== variant: [$v]" | tee -a gpurun_out/variants/out.txt
  TACEX_EXTRA_HIPCC_FLAGS="$v" python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants/out.txt
done
