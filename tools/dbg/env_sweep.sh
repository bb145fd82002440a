#!/bin/bash
# kernel times of one workload under a list of environment settings, on one box
# usage: tools/dbg/env_sweep.sh <tag> <workload> "VAR=a" "VAR=b" ...
TAG=${1:-r5z}; W=${2:-c3}; shift 2
cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
for E in "A=1" "$@" "A=2"; do
  env $E timeout 300 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$E', 'ms/step %.4f' % d['ms_per_step'], {a: round(v,4) for a,v in d['config']['kernel_ms'].items()}, 'tiles', d['config']['tiles'])"
done 2>&1 | tee $OUT/sweep.txt
