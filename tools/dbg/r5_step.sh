#!/bin/bash
# round 5: one memset per pass (cursor words in front of the look-back words), n_runs in the parameter block, parameter upload only when
# it changed — parity on the golden / keyed / hemi cases, then the step on C3 / C2 / hemi
TAG=${1:-r5t}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_parity_hemi.py tests/test_gpu_bedgraph.py tests/test_gpu_abi_client.py tests/test_gpu_loud_failures.py ${EXTRA_TESTS:-} -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
run() { # name, workload, extra bench args...
  local name=$1 wl=$2; shift 2
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --workload $wl "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); r=d["roofline"]; k=d["config"]["kernel_ms"]
    e=d["tiers"].get("end_to_end") or {}
    print("$name", "ms/step %.4f"%d["ms_per_step"], {a: round(v,4) for a,v in k.items()}, "outside kernels %.4f"%(d["ms_per_step"]-sum(k.values())), "e2e", e.get("ms"))
except Exception as e: print("$name", "ERR", e)
PY
}
run c3 c3 --skip-e2e
run c3_b c3 --skip-e2e
run c2 c2 --skip-e2e
run hemi hemi --skip-e2e
run c3_e2e c3
