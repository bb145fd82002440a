#!/bin/bash
# round-3 first GPU pass of the slot pipeline: A/B diff, parity suite (8 workers), C3 kernel times
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3a; mkdir -p $OUT
timeout 300 python tools/dbg/slotdiff.py > $OUT/slotdiff.log 2>&1; echo "slotdiff exit $?"; grep -c MISMATCH $OUT/slotdiff.log; grep -B0 -A9 "^DIFF" $OUT/slotdiff.log | head -60; tail -2 $OUT/slotdiff.log
timeout 500 python -m pytest tests -m gpu -q -n 8 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 25 $OUT/pytest.log | cut -c1-300
for P in slots tiles; do
  if [ $P = tiles ]; then export MKP_PIPELINE=tiles; else unset MKP_PIPELINE; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench_$P.json 2> $OUT/bench_$P.err; echo "bench $P exit $?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$P.json")); print("$P ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"])
except Exception as e: print("bench parse failed", e); print(open("$OUT/bench_$P.err").read()[-1500:])
PY
done
