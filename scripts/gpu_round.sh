#!/bin/bash
# One GPU-box round trip: parity tests, the driver's bench command, a rocprofv3 kernel-trace of the headline config.
# usage: scripts/gpu_round.sh <tag> [tests|bench|prof|pmc ...]   (default: tests bench prof)
TAG=${1:-r02}; shift
STEPS=${@:-tests bench prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for s in $STEPS; do
  case $s in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log ;;
    bench) timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --details-out $OUT/bench_details.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.err; head -c 1500 $OUT/bench.json ;;
    benchq) timeout 600 python bench.py --no-sweep --no-node-leg --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "benchq rc=$?"; head -c 1200 $OUT/bench_quick.json ;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-node-leg > $OUT/prof.log 2>&1; echo "prof rc=$?") ;;
    prof640) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof640 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --height 480 --width 640 --envs-per-gpu 1024 --sensors 1 --no-cpu-baseline --no-sweep --no-node-leg > $OUT/prof640.log 2>&1; echo "prof640 rc=$?"; tail -c 700 $OUT/prof640.log) ;;
    proffem) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proffem -- python3 $GRAFT_REPO_ROOT/scripts/fem_bench.py > $OUT/proffem.log 2>&1; echo "proffem rc=$?"; tail -12 $OUT/proffem.log) ;;
    pmc) (cd /tmp && export TMPDIR=/tmp
          timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-node-leg --no-roofline > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
          timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-node-leg --no-roofline > $OUT/pmc_write.log 2>&1; echo "pmc write rc=$?") ;;
    pmc640) (cd /tmp && export TMPDIR=/tmp
          for c in FETCH_SIZE WRITE_SIZE; do d=$(echo $c | tr A-Z a-z | sed 's/_size//')
            timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc640_$d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --height 480 --width 640 --envs-per-gpu 1024 --sensors 1 --no-cpu-baseline --no-sweep --no-node-leg --no-roofline > $OUT/pmc640_$d.log 2>&1; echo "pmc640 $c rc=$?"; done) ;;
    pmcfem) (cd /tmp && export TMPDIR=/tmp
          for c in FETCH_SIZE WRITE_SIZE; do d=$(echo $c | tr A-Z a-z | sed 's/_size//')
            timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcfem_$d -- python3 $GRAFT_REPO_ROOT/scripts/fem_bench.py > $OUT/pmcfem_$d.log 2>&1; echo "pmcfem $c rc=$?"; done) ;;
    timeline) (cd /tmp && export TMPDIR=/tmp
          timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_c4 -- python3 $GRAFT_REPO_ROOT/scripts/r06/c4_timeline.py > $OUT/c4_timeline_run.log 2>&1; echo "timeline c4 rc=$?"; grep "C4:" $OUT/c4_timeline_run.log
          python3 $GRAFT_REPO_ROOT/scripts/r06/timeline_of.py $OUT/prof_c4 fem_newton_lds_kernel 4 > $OUT/c4_step_timeline.txt; rm -rf $OUT/prof_c4
          python3 $GRAFT_REPO_ROOT/scripts/r06/timeline_of.py $OUT/prof taxim_stream_kernel 12 > $OUT/c3_step_timeline.txt) ;;
    pmcsq) (cd /tmp && export TMPDIR=/tmp
          timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-node-leg --no-roofline > $OUT/pmc_sq.log 2>&1; echo "pmc sq rc=$?") ;;
  esac
done
# keep the merged-back payload small: kernel stats + counter csvs only
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $OUT
