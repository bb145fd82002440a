#!/bin/bash
# the first-launch wait of a shard pass (MKP_TRACE_PLAN laps of the in-process end-to-end runs) under a list of environment settings
# usage: tools/dbg/r5_e2e_env.sh <tag> "VAR=a" "VAR=b VAR2=c" ...
TAG=${1:-r5ad}; shift
cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
i=0
for E in "A=1" "$@"; do
  i=$((i+1))
  env $E MKP_TRACE_PLAN=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "== $E"; grep -E "first event reached|kernels: sync|run: make_resident|output closed" $OUT/bench_$i.err | head -8 | cut -c1-120
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
for t in ['end_to_end','end_to_end_warm_context']:
    e=d['tiers'].get(t)
    if e: print(t,round(e['ms'],1))
PY
done 2>&1 | tee $OUT/env.txt
