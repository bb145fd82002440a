#!/bin/bash
# One GPU-box pass: parity tests, bench line (with its own --pmc traffic passes), rocprofv3 kernel-trace summary.
# Usage: tools/gpu_check.sh <tag> [pytest|nopytest|"pytest args"] [bench args...]
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt
PT=${2:-pytest}
shift; shift
if [ "$PT" = "pytest" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
  tail -5 $OUT/pytest.log
elif [ "$PT" != "nopytest" ]; then
  timeout 1500 python -m pytest $PT > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
  tail -15 $OUT/pytest.log
fi
timeout 1200 python bench.py --steps 20 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e "$@" > $OUT/bench_rocprof.json 2> $OUT/rocprof.err; echo "rocprof exit $?"
for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
cat $OUT/kernel_stats.csv 2>/dev/null | head -14
