#!/bin/bash
# round 5: persistent mkp_pileup_stream workgroups (next tile's chain prefetched by the last wave) and decode kernels on two streams —
# parity on the slot / hemi / keyed paths, then A/B on the C3 and hemi benches (env knobs select the old behaviour)
TAG=${1:-r5p}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_parity_fuzz.py tests/test_gpu_parity_hemi.py tests/test_gpu_bedgraph.py tests/test_gpu_sample_probs.py tests/test_gpu_scale.py -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e ${WL:+--workload $WL} > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", "ms/step %.4f"%d["ms_per_step"], {k: round(v,4) for k,v in d["config"]["kernel_ms"].items()})
except Exception as e: print("$name", "ERR", e)
PY
}
run c3_default A=1
run c3_default2 A=1
run c3_grid_tiles MKP_STREAM_GRID=0
run c3_grid_256 MKP_STREAM_GRID=256
run c3_grid_768 MKP_STREAM_GRID=768
run c3_no_overlap MKP_DECODE_OVERLAP=0
run c3_w4 MKP_PILEUP_WAVES=4
WL=hemi run hemi_default A=1
WL=hemi run hemi_no_overlap MKP_DECODE_OVERLAP=0
# full-size parity of the default build (sha256 against the oracle happens in the main bench; here: against the one-tile-per-workgroup order)
timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --cpu-sample region > $OUT/bench_full.json 2> $OUT/bench_full.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_full.json").read().strip().splitlines()[-1]); print("full", d["ms_per_step"], d["cpu_baseline"].get("bedmethyl_sha256_equal"), d["tiers"]["end_to_end"]["ms"])
PY
# HBM bytes of the stream kernel: the 64-VGPR build (scratch) against the 128-VGPR one
for V in 8 4; do MKP_PILEUP_WAVES=$V timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-e2e > $OUT/pmc_w$V.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/pmc_w$V.json").read().strip().splitlines()[-1]); r=d["roofline"]; print("waves $V", r["kernel"], r["avg_launch_ms"], "read", r["traffic_read"], "written", r["traffic_write"])
PY
done
