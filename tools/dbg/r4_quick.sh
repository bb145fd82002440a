#!/bin/bash
# round 4: quick GPU loop — the ingest tests, a fuzz slice, the C3 bench line and one traced CLI run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r4q}
exec > gpurun_out/$TAG.log 2>&1
set -x
timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_parity_fuzz.py -x -q -m gpu -k "fuzz_mixed and (12- or 19- or 15- or 16-)" 2>&1 | tail -8
export MKP_BENCH_DIR=/tmp
MKP_TRACE_PLAN=1 timeout 1500 python bench.py --steps 5 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
grep -v "mkpileup plan" gpurun_out/${TAG}_bench.err | cut -c1-250 | head -40
BAM=/tmp/mkp_c3_L64444167_N193000_x1_seed20.bam; FA=${BAM%.bam}.fa
MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $BAM /tmp/o_dev.bed --cpg --ref $FA -t 8 --stats 2> gpurun_out/${TAG}_trace_dev.txt
grep -E "mkpileup run|device ingest|total_ms|mkpileup ingest|threshold sampling|env overrides" gpurun_out/${TAG}_trace_dev.txt | cut -c1-250
MKP_HOST_INGEST=1 ./modkit_amd/csrc/mkpileup pileup $BAM /tmp/o_host.bed --cpg --ref $FA -t 8 --stats 2> gpurun_out/${TAG}_trace_host.txt
cmp /tmp/o_dev.bed /tmp/o_host.bed && echo DEV_EQUALS_HOST
