#!/bin/bash
# VALU instructions of the slot decoders per read, by part: SQ counters of a -DMKP_DEBUG build (tools/dbg/build_variant.sh debug "-DMKP_DEBUG")
# under MKP_DEBUG_SKIP = 0 | 64 (no CIGAR mapping) | 128 (no rank lookups / caller) | 256 (no sweep, no calls) | 512 (no slot loop) | sums
# usage: tools/dbg/r6_decode_valu.sh <tag> [skip ...]
TAG=${1:-dv}; shift; cd "$(dirname "$0")/../.." && export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp MKP_BENCH_DIR=/tmp MKP_LIB_PATH=$PWD/tools/dbg/lib/libmkpileup_${LIB:-debug}.so
for K in ${@:-0 64 128 192 256 512 768}; do
  echo "== skip $K" | tee -a $OUT/valu.txt
  MKP_DEBUG_SKIP=$K PASSES=1 bash tools/dbg/pmc_wide.sh $TAG/s$K 2>&1 | grep -E "mkp_decode|mkp_cover|failed" | cut -c1-330 | tee -a $OUT/valu.txt
done
