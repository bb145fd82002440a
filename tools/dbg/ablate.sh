#!/bin/bash
# Ablation of the accumulate kernel on the bench workload with a -DMKP_DEBUG build of the library (tools/dbg/lib/libmkpileup_debug.so:
# `make -C modkit_amd/csrc clean all CXXFLAGS="... -DMKP_DEBUG"`).  MKP_DEBUG_SKIP bits: 1 depth walk, 2 events, 4 row emission, 8 SEQ phase.
# Usage: [WORKLOAD=c2] tools/dbg/ablate.sh <tag> <skip>...
TAG=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export MKP_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/lib/libmkpileup_debug.so
for K in "$@"; do
  MKP_DEBUG_SKIP=$K timeout 300 python bench.py --workload ${WORKLOAD:-c3} --steps 20 --warmup 3 --skip-e2e --no-pmc > $OUT/bench_$K.json 2> $OUT/bench_$K.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$K.json"))
print("skip $K", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "tiles", d["config"]["tiles"], "rows", d["config"]["rows_per_step"])
PY
done
