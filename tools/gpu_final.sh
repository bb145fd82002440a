#!/bin/bash
# One GPU-box pass for the round's evidence: (optionally) the parity suite as the driver runs it, the default bench line (PMC traffic passes,
# CPU baselines, seam tiers), the c2 / hemi lines, rocprofv3 kernel-trace summaries of the timed step (c3 / c2 / hemi) and of one whole
# `mkpileup pileup` run (the device-ingest kernels), SQ counters of the c3 step kernels and of the inflate kernels.
# Usage: [SKIP_PYTEST=1] [EXTRA_WORKLOADS=""] [SKIP_N2=1] [SKIP_INFLATE_PMC=1] tools/gpu_final.sh <tag>     -> gpurun_out/<tag>/
set -u
TAG=${1:-r05}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
if [ -z "${SKIP_PYTEST:-}" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
fi
export MKP_BENCH_DIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/c3_bench.json 2> $OUT/c3_bench.err; echo "bench c3 exit $?"; tail -2 $OUT/c3_bench.err | cut -c1-300
for W in ${EXTRA_WORKLOADS-c2 hemi}; do timeout 600 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"; done
cd /tmp
for W in c3 ${EXTRA_WORKLOADS-c2 hemi}; do
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e > $OUT/${W}_bench_rocprof.json 2> $OUT/${W}_rocprof.err; echo "rocprof $W exit $?"
  for f in $(find /tmp/prof_$W -name '*kernel_stats.csv'); do cp $f $OUT/${W}_kernel_stats.csv; done
  head -8 $OUT/${W}_kernel_stats.csv | cut -c1-160
done
# one whole run of the subcommand on the bench BAM (the c3 bench left it under $MKP_BENCH_DIR): the ingest kernels next to the step kernels
P=$(ls /tmp/mkp_c3_*.bam 2>/dev/null | head -1); P=${P%.bam}
if [ -n "$P" ]; then
  rm -rf /tmp/prof_cli
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli -o cli -- $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/ingest_cli.err; echo "rocprof cli exit $?"
  for f in $(find /tmp/prof_cli -name '*kernel_stats.csv'); do cp $f $OUT/ingest_kernel_stats.csv; done
  head -14 $OUT/ingest_kernel_stats.csv | cut -c1-160; grep -E "ingest|total_ms|MKP_" $OUT/ingest_cli.err | cut -c1-300 | head
fi
# the 2-rank form of the bench on this box's one GPU (gloo): the sharded path end to end, its parity flags
[ -n "${SKIP_N2:-}" ] || ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo ) > $OUT/n2_gloo_bench.json 2> $OUT/n2_gloo_bench.err; echo "n2 gloo exit $?"
cd $GRAFT_REPO_ROOT; PASSES="1 2" bash tools/dbg/pmc_wide.sh $TAG/sq > /dev/null 2>&1; cat $OUT/sq/pmc.txt | cut -c1-400
[ -n "${SKIP_INFLATE_PMC:-}" ] || KERNELS=wave4 bash tools/dbg/pmc_inflate.sh $TAG/sqi > /dev/null 2>&1; [ -n "${SKIP_INFLATE_PMC:-}" ] || cat $OUT/sqi/sq_inflate.txt | cut -c1-500
