#!/bin/bash
# shards ingested side by side: which inflate kernel, how many in flight (c4 / c5 scale models); the RCCL one-rank test
TAG=${1:-r4k}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k rccl 2>&1 | grep -v "^$" | tail -30 | cut -c1-260 > $OUT/rccl.log; grep -E "passed|failed|Error|WARN" $OUT/rccl.log | head
export MKP_BENCH_DIR=/tmp
for W in c4 c5; do for CFG in "thread2 4" "thread2 8" "wave2 8" "thread2 12"; do
  set -- $CFG
  MKP_INFLATE_KERNEL=$1 MKP_AHEAD_WORKERS=$2 timeout 600 python bench.py --workload $W --steps 1 --warmup 0 --no-pmc --no-cpu-baseline > $OUT/${W}_$1_$2.json 2> /dev/null
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/${W}_$1_$2.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
print("$W $1 nw=$2 e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()})
PY
done; done
