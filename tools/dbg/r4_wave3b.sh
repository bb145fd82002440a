#!/bin/bash
# wave3 as the shard-sized default: ingest tests, c5 with the CPU baseline's sha256
TAG=${1:-r4z2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-300
export MKP_BENCH_DIR=/tmp
( timeout 400 python bench.py --workload c5 --steps 1 --warmup 0 --no-pmc ) > $OUT/c5_bench.json 2> $OUT/c5_bench.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/c5_bench.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
    print("c5 e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "sha equal:", (d.get("cpu_baseline") or {}).get("bedmethyl_sha256_equal"))
except Exception as ex: print("c5 failed", ex, open("$OUT/c5_bench.err").read()[-400:])
PY
