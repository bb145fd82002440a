#!/bin/bash
# round 5: the scale models after the staged-copy change (no pageable buffer reaches the runtime): parity subset, then C4 / C5 1/10 end to end
TAG=${1:-r5c4}; shift; WLS=${@:-c4}
cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
if [ -z "${SKIP_PYTEST:-}" ]; then ( timeout 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_extract.py tests/test_gpu_inflate.py tests/test_gpu_sample_probs.py tests/test_gpu_summary.py -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log; fi
for W in $WLS; do
  timeout 900 python bench.py --workload $W --steps 1 --warmup 0 --no-pmc > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/${W}_bench.json").read().strip().splitlines()[-1]); e=d["tiers"]["end_to_end"]
print("$W", "e2e ms", round(e["ms"],1), "shards", e.get("shards"), {k: round(v,1) for k,v in e["stages_ms"].items()}, "sha", d.get("cpu_baseline",{}).get("bedmethyl_sha256_equal"), d.get("cpu_baseline",{}).get("sample","")[:60])
f=d["tiers"].get("full_data_threshold_run")
if f: print("   -f 1.0:", {k: (round(v,1) if isinstance(v,(int,float)) else v) for k,v in f.items() if k in ("ms","threshold_ms","total_ms")})
PY
done
