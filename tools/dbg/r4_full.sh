#!/bin/bash
# round 4: the whole GPU suite + the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r4f}
exec > gpurun_out/$TAG.log 2>&1
set -x
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -30
export MKP_BENCH_DIR=/tmp
MKP_TRACE_PLAN=1 timeout 1500 python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
grep -v "mkpileup plan" gpurun_out/${TAG}_bench.err | cut -c1-250 | head -30
