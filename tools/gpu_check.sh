#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel-trace summary, HBM-traffic PMC passes.
# Usage: tools/gpu_check.sh <tag> [pytest|nopytest] [pmc|nopmc]
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "${2:-pytest}" = "pytest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_rocprof.json 2> $OUT/rocprof.err; echo "rocprof exit $?"
find /tmp/prof_$TAG -type f | head -20
for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
cat $OUT/kernel_stats.csv 2>/dev/null | head -12
if [ "${3:-pmc}" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err; echo "pmc $C exit $?"
    for f in $(find /tmp/pmc_${TAG}_$C -name '*counter_collection.csv'); do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $C > $OUT/pmc_$C.txt; done
    cat $OUT/pmc_$C.txt
  done
fi
