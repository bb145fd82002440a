#!/bin/bash
# One GPU-box pass for the round's evidence: the parity suite as the driver runs it (serial), the default bench line (PMC traffic passes,
# CPU baselines, seam tier), rocprofv3 kernel-trace summaries of the c3 / c2 / hemi workloads, SQ counters of the c3 kernels.
# Usage: tools/gpu_final.sh <tag>     -> gpurun_out/<tag>/
set -u
TAG=${1:-r03}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/c3_bench.json 2> $OUT/c3_bench.err; echo "bench c3 exit $?"; tail -2 $OUT/c3_bench.err | cut -c1-300
for W in c2 hemi; do timeout 600 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"; done
cd /tmp
for W in c3 c2 hemi; do
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e > $OUT/${W}_bench_rocprof.json 2> $OUT/${W}_rocprof.err; echo "rocprof $W exit $?"
  for f in $(find /tmp/prof_$W -name '*kernel_stats.csv'); do cp $f $OUT/${W}_kernel_stats.csv; done
  head -8 $OUT/${W}_kernel_stats.csv | cut -c1-160
done
cd $GRAFT_REPO_ROOT; PASSES="1 2" bash tools/dbg/pmc_wide.sh $TAG/sq > /dev/null 2>&1; cat $OUT/sq/pmc.txt | cut -c1-400
