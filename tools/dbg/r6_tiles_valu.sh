#!/bin/bash
# VALU instructions of mkp_pileup_tiles (C2) by part: SQ counters of the -DMKP_DEBUG build under MKP_DEBUG_SKIP = 0 | 1 (no depth walk) | 2 (no events) | 4 (no rows) | 8 (no SEQ phase) | sums
TAG=${1:-tv}; shift; cd "$(dirname "$0")/../.." && export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp MKP_BENCH_DIR=/tmp MKP_LIB_PATH=$PWD/tools/dbg/lib/libmkpileup_debug.so BENCH_ARGS="--workload ${WORKLOAD:-c2}"
for K in ${@:-0 1 2 4 8 15}; do
  echo "== skip $K" | tee -a $OUT/valu.txt
  MKP_DEBUG_SKIP=$K PASSES=1 bash tools/dbg/pmc_wide.sh $TAG/s$K 2>&1 | grep -E "mkp_pileup|failed" | cut -c1-330 | tee -a $OUT/valu.txt
done
