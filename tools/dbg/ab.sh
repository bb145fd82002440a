#!/bin/bash
# A/B of several builds of the library on ONE box, alternating runs: `tree` = the in-tree build, any other name = tools/dbg/lib/libmkpileup_<name>.so
# (tools/dbg/build_variant.sh), used through MKP_LIB_PATH.  Kernel times of the timed step per build and round.
# usage: tools/dbg/ab.sh <tag> "<name> <name> ..." [workloads...]        WORKLOADS default: c3;  ROUNDS=2
TAG=${1:-ab}; NAMES=${2:-tree}; shift 2; WLS=${@:-c3}
cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
for W in $WLS; do for R in $(seq 1 ${ROUNDS:-2}); do
  for L in $NAMES; do
    if [ $L = tree ]; then unset MKP_LIB_PATH; else export MKP_LIB_PATH=$PWD/tools/dbg/lib/libmkpileup_$L.so; fi
    timeout 300 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e 2>$OUT/err_${W}_$L.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '%-10s' % '$L', 'ms/step %.4f' % d['ms_per_step'], {a: round(v,4) for a,v in d['config']['kernel_ms'].items()}, 'rows_checked', (d['config'].get('timed_rows_checked') or {}).get('equal'))
except Exception as e:
    print('$W', '$L', 'FAILED', e)"
  done
done; done 2>&1 | tee $OUT/ab.txt
