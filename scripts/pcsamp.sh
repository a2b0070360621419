#!/bin/bash
# PC sampling of the streaming tail (rocprofv3 beta feature): where do the waves of taxim_stream_kernel sit?
O=$GRAFT_REPO_ROOT/gpurun_out/pcs; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for method in stochastic host_trap; do
  unit=cycles; iv=65536
  [ $method = host_trap ] && unit=time && iv=50
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $iv --kernel-trace --output-format csv -d $O/$method -- python3 $GRAFT_REPO_ROOT/scripts/tail_bench.py 1024 1 > $O/$method.log 2>&1
  echo "$method rc=$?"; tail -3 $O/$method.log | cut -c1-200
  ls -la $O/$method/*/ 2>/dev/null | head
done
find $O -name "*.db" -delete
du -sh $O
