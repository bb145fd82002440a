#!/bin/bash
# quick GPU pass: a parity subset, then the bench line of both workloads without CPU baseline / PMC (kernel times only)
# Usage: tools/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-q}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
K=${2:-"golden or fuzz or scaled"}
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
for W in c3 c2; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err; echo "bench $W exit $?"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$W.json"))
print("$W", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"], "frac %.4f"%d["roofline"]["frac"], "e2e_ms %.0f"%d["tiers"]["end_to_end"]["ms"], d["tiers"]["end_to_end"]["stages_ms"])
PY
done
