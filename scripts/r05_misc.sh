#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05misc}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fem_gpu.py tests/test_fem_physics_gpu.py -q -m gpu -x 2>&1 | tail -6 | tee $OUT/fem_tests.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/scripts/fem_bench.py > $OUT/prof.log 2>&1)
grep "FemGelpad scene" $OUT/prof.log
python - <<P
import csv,glob
f=glob.glob('$OUT/prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'assemble' in r['Name'] or 'newton_lds' in r['Name']:
        print(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3)
P
find $OUT -name "*.db" -delete 2>/dev/null
