#!/bin/bash
# round 5: (1) A/B of mkp_pileup_stream's visit batch (MKP_STREAM_BATCH=1|2|4) with the re-ordered prologue, parity subset first;
# (2) parity at the scale the C4 / C5 lines are quoted on: sha256 of the WHOLE 1/10 scale-model outputs against the oracle
TAG=${1:-r5g}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_parity_fuzz.py tests/test_gpu_bedgraph.py -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for B in 4 2 1 4 2 1; do
  MKP_STREAM_BATCH=$B timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc --no-cpu-baseline > $OUT/ab_b$B.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/ab_b$B.json')); print('batch $B', 'ms/step %.3f' % d['ms_per_step'], d['config']['kernel_ms'])"
done
for W in c4 c5; do
  ( time timeout 1500 python bench.py --workload $W --steps 2 --warmup 1 --no-pmc --cpu-whole ) > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"; tail -3 $OUT/${W}_bench.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${W}_bench.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]; cb=d.get("cpu_baseline") or {}
    print("$W", "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "shards", e.get("shards"))
    print("   cpu", {k: cb.get(k) for k in ("value","cores","sample","bedmethyl_sha256_equal","rows")}, (cb.get("end_to_end") or {}).get("total_s"))
except Exception as ex: print("$W parse failed", ex)
PY
done
