#!/bin/bash
# end-to-end tiers of the default bench under several builds of the library on ONE box (tools/dbg/lib/libmkpileup_<name>.so; "tree" = the in-tree build)
# usage: tools/dbg/e2e_ab.sh <tag> "<name> ..." [rounds]
TAG=${1:-e2eab}; LIBS=${2:-"base tree"}; ROUNDS=${3:-2}
cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
for r in $(seq $ROUNDS); do for L in $LIBS; do
  if [ "$L" = tree ]; then unset MKP_LIB_PATH; else export MKP_LIB_PATH=$PWD/tools/dbg/lib/libmkpileup_$L.so; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-pmc --no-cpu-baseline 2> $OUT/err_$L.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tiers']
for k in ('end_to_end','end_to_end_warm_context'):
    s=t[k]['stages_ms']; print('$L', k, 'ms %.1f' % t[k]['ms'], {a: round(v,1) for a,v in s.items() if v >= 1.0})
print('$L', 'ingest host_ms', d['roofline']['ingest'].get('host_ms'), 'inflate ms', round(d['roofline']['ingest']['avg_launch_ms'],1))"
done; done 2>&1 | tee $OUT/e2e_ab.txt
