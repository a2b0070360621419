#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
TACEX_LIB_TAG=clk8 TACEX_STREAM_LDS_PAD=24576 python scripts/stream_clock8.py 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
TACEX_LIB_TAG=clk8 python scripts/stream_clock8.py 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
for i in 1 2; do python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-330 | tee -a $O/out.txt; done
timeout 900 python -m pytest tests/test_taxim_gpu.py tests/test_edge_cases_gpu.py tests/test_sensor_gpu.py tests/test_sensor_configs_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee -a $O/out.txt
