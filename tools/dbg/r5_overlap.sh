#!/bin/bash
# round 5: inflate capped at 12 waves per CU + context streams at the highest priority — do the small kernels run beside an ingest now?
# C4 / C5 scale models (default mode, and C4 with -f 1.0), C3 end to end, the sampling rounds' times from --stats; parity subset first.
TAG=${1:-r5h}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_ingest.py tests/test_gpu_parity_golden.py -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for k in wave4; do for n in 4000 0; do if [ $n = 0 ]; then unset INFLATE_BLOCKS; else export INFLATE_BLOCKS=$n; fi
  echo "kernel $k blocks $n: $(python tools/dbg/inflate_bench.py 2>&1 | tail -1 | cut -c1-330)"; done; done; unset INFLATE_BLOCKS
for W in c4 c5 c3; do
  ( time timeout 900 python bench.py --workload $W --steps 2 --warmup 1 --no-pmc --no-cpu-baseline ) > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${W}_bench.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
    print("$W", "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "shards", e.get("shards"), json.dumps(d["config"].get("full_data_threshold_run")))
    w=d["tiers"].get("end_to_end_warm_context")
    if w: print("   warm %.0f"%w["ms"], {k:round(v) for k,v in w["stages_ms"].items()})
except Exception as ex: print("$W parse failed", ex)
PY
done
P=/tmp/mkp_c4_g0.1_seed40.bam; F=${P%.bam}.fa
for i in 1 2; do MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P /tmp/o_c4.bed --preset traditional --ref $F -t 8 --stats 2> $OUT/c4_cli_trace.err > /dev/null; done
grep -E "threshold sampling|total_ms|resident_sampling" $OUT/c4_cli_trace.err | cut -c1-330
grep "mkpileup run\]" $OUT/c4_cli_trace.err | sed -n '1,60p' | cut -c1-110
