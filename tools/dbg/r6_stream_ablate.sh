#!/bin/bash
# where mkp_pileup_stream's time goes (round 6 kernel): MKP_DEBUG_SKIP on the -DMKP_DEBUG build (tools/dbg/build_debug.sh) —
# 1024 no visits, 2048 no rows (look-back word still published), 3072 both, 4096 no look-back (rows in completion order)
TAG=${1:-r6a}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
export MKP_LIB_PATH=$PWD/tools/dbg/lib/libmkpileup_debug.so
for K in 0 1024 2048 3072 4096 0; do
  MKP_DEBUG_SKIP=$K timeout 300 python bench.py --workload ${WORKLOAD:-c3} --steps 20 --warmup 3 --skip-e2e --no-pmc --no-cpu-baseline 2>$OUT/ab_$K.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip %-5s' % '$K', 'pileup %.4f decode %.4f step %.3f' % (d['config']['kernel_ms']['pileup'], d['config']['kernel_ms']['decode'], d['ms_per_step']))
except Exception as e: print('skip $K failed', e)"
done 2>&1 | tee $OUT/stream_ablation.txt
