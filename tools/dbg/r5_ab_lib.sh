#!/bin/bash
# A/B of two builds of the library on one box: the in-tree one and another through MKP_LIB_PATH (tools/dbg/lib/libmkpileup_<name>.so)
# usage: tools/dbg/r5_ab_lib.sh <tag> <name> [workloads...]
TAG=${1:-r5v}; NAME=${2:-r05}; shift 2; WLS=${@:-c2 c3}
cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
V=$PWD/tools/dbg/lib/libmkpileup_$NAME.so
for W in $WLS; do for R in 1 2; do
  for L in tree $NAME; do
    if [ $L = tree ]; then unset MKP_LIB_PATH; else export MKP_LIB_PATH=$V; fi
    timeout 300 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$L', 'ms/step %.4f' % d['ms_per_step'], {a: round(v,4) for a,v in d['config']['kernel_ms'].items()})"
  done
done; done 2>&1 | tee $OUT/ab.txt
