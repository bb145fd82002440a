#!/bin/bash
# round 4: quick GPU loop — the tests the day's changes touch, the C3 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r4q}
exec > gpurun_out/$TAG.log 2>&1
set -x
timeout 2000 python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_abi_client.py tests/test_gpu_loud_failures.py tests/test_gpu_ingest.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_parity_fuzz.py -x -q -m gpu -k "fuzz_mixed and (12- or 19- or 15- or 4- or 8-)" 2>&1 | tail -8
export MKP_BENCH_DIR=/tmp
MKP_TRACE_PLAN=1 timeout 1500 python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
grep -v "mkpileup plan" gpurun_out/${TAG}_bench.err | cut -c1-250 | head -40
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("ms_per_step", d["ms_per_step"], d["config"]["kernel_ms"])
for k in ("end_to_end","end_to_end_warm_context"):
    print(k, round(d["tiers"][k]["ms"],1), {a:round(b,1) for a,b in d["tiers"][k]["stages_ms"].items()})
print("seam", d["tiers"].get("seam_per_interval",{}).get("rows_per_s_api"), {k:v.get("rows_per_s_api") for k,v in d["tiers"].get("seam_batch",{}).items()})
PY
