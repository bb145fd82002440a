#!/bin/bash
# experiments: accumulate-kernel time of the bench workload against the tile span.  Usage: tools/dbg/tile_sweep.sh <tag> <span>...
TAG=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for T in "$@"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc --tile $T > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$T.json"))
print("tile $T", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"])
PY
done
