#!/bin/bash
# round 5: where an end-to-end C3 run spends its wall time — MKP_TRACE_PLAN timeline of the in-process runs of the bench (fresh and warm context)
TAG=${1:-r5u}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
MKP_TRACE_PLAN=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench.json 2> $OUT/bench.err
grep -E "^\[mkpileup" $OUT/bench.err | cut -c1-200 > $OUT/trace.txt; wc -l $OUT/trace.txt
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); k=d["config"]["kernel_ms"]
print("ms/step %.4f"%d["ms_per_step"], {a: round(v,4) for a,v in k.items()}, "outside kernels %.4f"%(d["ms_per_step"]-sum(k.values())))
for t in ['end_to_end','end_to_end_warm_context']:
    e=d['tiers'].get(t)
    if e: print(t,round(e['ms'],1),{a:round(v,1) for a,v in e['stages_ms'].items()})
PY
timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d['config']['kernel_ms'])"
