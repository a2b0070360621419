#!/bin/bash
# round-3 A/B: conflict-free H-pass window map of the MFMA band kernels (parity + stage times + LDS conflict counters)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests/test_taxim_gpu.py tests/test_edge_cases_gpu.py tests/test_sensor_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
for i in 1 2; do python scripts/tail_bench.py 1024 1 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $O/out.txt; done
python scripts/tail_bench.py 1024 1 480 640 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $O/out.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc -- python3 $GRAFT_REPO_ROOT/scripts/tail_bench.py 1024 1 > $O/pmc.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03n"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "blur_mfma" in k: acc[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    m={c: sum(x)/len(x) for c,x in v.items()}
    print(k, "conflict ratio %.3f" % (m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"]), {c: round(x) for c,x in m.items()})
PY
find $O -name "*.db" -delete
