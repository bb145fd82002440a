#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3h; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_scale.py -q -x -k "genome_scale or two_processes" ) > $OUT/pytest_scale.log 2>&1; echo "pytest scale exit $?"; tail -n 8 $OUT/pytest_scale.log | cut -c1-400
( time timeout 900 python bench.py --steps 20 --warmup 3 ) > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench c3 exit $?"; tail -3 $OUT/bench_c3.err | cut -c1-600
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_c3.json").readline()); print("c3 ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"]); print(json.dumps(d["roofline"])[:1500]); print(json.dumps(d["tiers"].get("seam_per_interval"))); print(json.dumps(d.get("cpu_baseline",{}).get("matched_host_threads"))[:900]); print("sha equal", d.get("cpu_baseline",{}).get("bedmethyl_sha256_equal"), "e2e ms", d["tiers"]["end_to_end"]["ms"])
except Exception as e: print("c3 parse failed", e)
PY
( time timeout 600 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 exit $?"; tail -2 $OUT/bench_c2.err | cut -c1-400
for W in c4 c5; do
( time timeout 900 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-pmc ) > $OUT/bench_$W.json 2> $OUT/bench_$W.err; echo "bench $W exit $?"; tail -4 $OUT/bench_$W.err | cut -c1-500
done
python - <<PY
import json
for w in ("c2","c4","c5"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%w).readline()); print(w, "ms/step %.3f"%d["ms_per_step"], "value %.3g"%d["value"], d["config"]["kernel_ms"], "e2e ms %.0f"%d["tiers"]["end_to_end"]["ms"], d["tiers"]["end_to_end"]["stages_ms"], "shards", d["tiers"]["end_to_end"]["shards"])
    except Exception as e: print(w, "parse failed", e)
PY
