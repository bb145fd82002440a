#!/bin/bash
# experiments: the bench workload's kernel times with alternative builds of the library, back to back on one box.
# Usage: tools/dbg/ab_lib.sh <tag> <lib.so|default>...
TAG=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for L in "$@" "$@"; do
  if [ "$L" = default ]; then unset MKP_LIB_PATH; else export MKP_LIB_PATH=$GRAFT_REPO_ROOT/$L; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("$L", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
