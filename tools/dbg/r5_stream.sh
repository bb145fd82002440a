#!/bin/bash
# round 5: mkp_pileup_stream with ticket-ordered tiles, four visits in flight per wave, wave-parallel look-back — parity, then kernel times on C3
TAG=${1:-r5e}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_parity_fuzz.py tests/test_gpu_bedgraph.py tests/test_gpu_loud_failures.py tests/test_gpu_inflate.py tests/test_gpu_abi_client.py -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for W in c3; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err; echo "bench $W exit $?"; tail -2 $OUT/bench_$W.err | cut -c1-300
  python - <<PY
import json
d=json.load(open("$OUT/bench_$W.json"))
print("$W", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"], "frac %.4f"%d["roofline"]["frac"], "e2e_ms %.0f"%d["tiers"]["end_to_end"]["ms"], {k: round(v) for k, v in d["tiers"]["end_to_end"]["stages_ms"].items()})
print("   warm", {k: round(v) for k, v in d["tiers"]["end_to_end_warm_context"]["stages_ms"].items()}, round(d["tiers"]["end_to_end_warm_context"]["ms"]))
print("   ingest", json.dumps(d["roofline"]["ingest"])[:500])
PY
done
MKP_PILEUP_WAVES=4 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e > $OUT/bench_c3_w4.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_c3_w4.json')); print('w4 build', d['ms_per_step'], d['config']['kernel_ms'])"
