#!/bin/bash
# gpurun_out/<tag> (scripts/gpu_round.sh <tag> tests bench prof pmc proffem pmcfem prof640 pmc640 pmcsq) -> profiles/r06_*
# usage: scripts/r06/collect_profiles.sh <tag> <commit>
T=gpurun_out/${1:-r06}; C=${2:-$(git rev-parse --short HEAD)}; P=profiles
cp $T/bench.json $P/r06_bench_n1.json
cp $T/bench_details.json $P/r06_bench_details_n1.json
cp $(ls $T/prof/*/*kernel_stats.csv | head -1) $P/r06_c3_kernel_stats.csv
cp $(ls $T/prof640/*/*kernel_stats.csv | head -1) $P/r06_c5_640x480_kernel_stats.csv
cp $(ls $T/proffem/*/*kernel_stats.csv | head -1) $P/r06_fem_kernel_stats.csv
grep -v "rocprofv3\|^[EW]2026" $T/proffem.log > $P/r06_fem_bench.log
python scripts/make_pmc_traffic.py $T/pmc_fetch $T/pmc_write $P/pmc_traffic_r06.json 0 240 320 $C 2048
python scripts/make_pmc_traffic.py $T/pmc640_fetch $T/pmc640_write $P/pmc_traffic_r06_640x480.json 0 480 640 $C 1024
python scripts/make_pmc_traffic_fem.py $T/pmcfem_fetch $T/pmcfem_write $T/proffem $P/pmc_traffic_r06_fem.json $C
for t in c3 c4; do [ -s $T/${t}_step_timeline.txt ] && sed "s#^\# .*kernel_trace.csv#\# one steady step at commit $C#" $T/${t}_step_timeline.txt > $P/r06_${t}_step_timeline.txt; done
python scripts/pmc_sq_report.py $T/pmc_sq $P/r06_pmc_sq_summary.md $C
ls -la $P/r06_* $P/pmc_traffic_r06*
