#!/bin/bash
# ingest-ahead with several shards in flight: the rccl one-rank test in full, scale/fuzz tests, c4 / c5 with traces
TAG=${1:-r4i}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k rccl 2>&1 | tail -60 > $OUT/pytest_rccl.log; tail -30 $OUT/pytest_rccl.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_ingest.py -m gpu -q --deselect tests/test_gpu_scale.py::test_histogram_allreduce_through_rccl_on_one_rank 2>&1 | tail -5 > $OUT/pytest_a.log; cat $OUT/pytest_a.log
timeout 900 python -m pytest tests/ -m gpu -x -q -k "fuzz or bed" 2>&1 | tail -5 > $OUT/pytest_b.log; cat $OUT/pytest_b.log
export MKP_BENCH_DIR=/tmp
for W in c4 c5; do
  MKP_TRACE_PLAN=1 timeout 900 python bench.py --workload $W --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"
  grep -E "mkpileup ingest|threshold|ahead" $OUT/${W}_bench.err | cut -c1-330 | tail -32
  for NW in 1 8; do MKP_AHEAD_WORKERS=$NW timeout 600 python bench.py --workload $W --steps 1 --warmup 0 --no-pmc --no-cpu-baseline > $OUT/${W}_nw$NW.json 2> /dev/null; done
done
python - <<PY
import json
for w in ("c4","c5","c4_nw1","c4_nw2","c4_nw8","c5_nw1","c5_nw2","c5_nw8"):
    try:
        f = "$OUT/%s_bench.json"%w if "_" not in w else "$OUT/%s.json"%w
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
        print(w, "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "shards", e.get("shards"))
    except Exception as ex: print(w, "parse failed", ex)
PY
