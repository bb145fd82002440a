#!/bin/bash
# round 6 probe: SQ counters of the c3 step kernels, stream-tile sweep, C2 / hemi kernel times and the C2 ablations of mkp_pileup_tiles
TAG=${1:-r6p}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
bash tools/dbg/env_sweep.sh $TAG/tile c3 MKP_STREAM_TILE=448 MKP_STREAM_TILE=640 MKP_STREAM_TILE=768 MKP_STREAM_TILE=896 2>&1 | tail -8
PASSES="1 2" bash tools/dbg/pmc_wide.sh $TAG/sq 2>&1 | cut -c1-420
for W in c2 hemi; do timeout 300 python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', 'ms/step %.4f' % d['ms_per_step'], {a: round(v,4) for a,v in d['config']['kernel_ms'].items()}, 'tiles', d['config']['tiles'], 'rows', d['config']['rows_per_step'])"; done
WORKLOAD=c2 bash tools/dbg/ablate.sh $TAG/abl 0 1 2 4 8 7 2>&1 | tail -8
