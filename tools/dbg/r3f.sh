#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r3f}; mkdir -p $OUT
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e > $OUT/bench_$1.json 2> $OUT/bench_$1.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$1.json")); print("$1 ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"])
except Exception as e: print("$1 bench parse failed", e); print(open("$OUT/bench_$1.err").read()[-800:])
PY
}
for V in ${VARIANTS:-0 1}; do export MKP_SLOT_VARIANT=$V; run v$V; done
unset MKP_SLOT_VARIANT
timeout 300 python tools/dbg/slotdiff.py > $OUT/slotdiff.log 2>&1; echo "slotdiff exit $?"; grep -B0 -A9 "^DIFF" $OUT/slotdiff.log | head -40; tail -1 $OUT/slotdiff.log
timeout 500 python -m pytest tests -m gpu -q -n 8 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 6 $OUT/pytest.log | cut -c1-300
