#!/bin/bash
# fused slot decoder variants (MKP_SLOT_VARIANT) and stream tile sizes on C3; A/B diff + parity suite with variant 1
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3b; mkdir -p $OUT
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e > $OUT/bench_$1.json 2> $OUT/bench_$1.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$1.json")); print("$1 ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"])
except Exception as e: print("$1 bench parse failed", e); print(open("$OUT/bench_$1.err").read()[-800:])
PY
}
for V in 0 1 2 3 4 5; do export MKP_SLOT_VARIANT=$V; run v$V; done
export MKP_SLOT_VARIANT=0
for T in 320 448 960; do export MKP_STREAM_TILE=$T; run t$T; done
unset MKP_STREAM_TILE
export MKP_SLOT_VARIANT=1
timeout 300 python tools/dbg/slotdiff.py > $OUT/slotdiff.log 2>&1; echo "slotdiff(v1) exit $?"; grep -B0 -A9 "^DIFF" $OUT/slotdiff.log | head -40; tail -1 $OUT/slotdiff.log
timeout 500 python -m pytest tests -m gpu -q -n 8 > $OUT/pytest.log 2>&1; echo "pytest(v1) exit $?"; tail -n 6 $OUT/pytest.log | cut -c1-300
