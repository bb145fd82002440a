#!/bin/bash
# ablations of mkp_decode_slots* (debug build, MKP_DEBUG_SKIP: 64 no CIGAR mapping, 128 no calls, 256 no sweep/calls, 512 no slot loop) + SQ counters
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/ablate_slots; mkdir -p $OUT
export MKP_SLOT_VARIANT=2
export MKP_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/lib/libmkpileup_debug.so
for K in 0 64 128 192 256 512 768; do
  MKP_DEBUG_SKIP=$K timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc --no-cpu-baseline > $OUT/bench_$K.json 2> $OUT/bench_$K.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$K.json")); print("skip $K", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"])
except Exception as e: print("skip $K failed", open("$OUT/bench_$K.err").read()[-400:])
PY
done
unset MKP_LIB_PATH
PASSES="1 2" bash tools/dbg/pmc_wide.sh r3d_pmc 2>&1 | cut -c1-600
