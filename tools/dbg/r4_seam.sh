#!/bin/bash
# the file seam (mkp_process_region through the device ingest): tests + the C3 seam tiers
TAG=${1:-r4s}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_abi_client.py tests/test_gpu_scale.py -m gpu -x -q -k "file_seam or full_size_properties or batch_seam_on" 2>&1 | tail -8 | cut -c1-400
export MKP_BENCH_DIR=/tmp
timeout 900 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/c3.json") if l.startswith("{")][-1]); t=d["tiers"]
print("e2e", round(t["end_to_end"]["ms"]), round(t["end_to_end_warm_context"]["ms"]))
print("seam per interval", t["seam_per_interval"].get("rows_per_s_api"), {k:v.get("rows_per_s_api") for k,v in t["seam_batch"].items()})
print("seam file", {k:t["seam_file"].get(k) for k in ("rows","rows_per_s_api","api_s","process_wall_s","error")})
PY
