#!/bin/bash
# round 4: device ingest on the GPU — parity tests, the C3 end-to-end run (in process and CLI traces), inflate kernel ablations
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r4c}
exec > gpurun_out/$TAG.log 2>&1
set -x
python -c "import __graft_entry__ as g; g.smoke()" || echo SMOKE_FAILED
timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_inflate.py tests/test_gpu_parity_golden.py -x -q -m gpu 2>&1 | tail -25
export MKP_BENCH_DIR=/tmp
MKP_TRACE_PLAN=1 timeout 1500 python bench.py --steps 5 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
grep -v "mkpileup plan" gpurun_out/${TAG}_bench.err | head -60
BAM=$(ls /tmp/mkp_c3_*seed20.bam | head -1); FA=${BAM%.bam}.fa
for k in thread2 thread; do
  MKP_INFLATE_KERNEL=$k MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $BAM /tmp/o_$k.bed --cpg --ref $FA -t 8 --stats 2> gpurun_out/${TAG}_trace_$k.txt
  grep -E "device ingest|total_ms|mkpileup ingest" gpurun_out/${TAG}_trace_$k.txt
done
MKP_HOST_INGEST=1 ./modkit_amd/csrc/mkpileup pileup $BAM /tmp/o_host.bed --cpg --ref $FA -t 8 --stats 2> gpurun_out/${TAG}_trace_host.txt
tail -1 gpurun_out/${TAG}_trace_host.txt
for k in thread2 thread; do cmp /tmp/o_$k.bed /tmp/o_host.bed && echo ${k}_EQUALS_HOST; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMKP_INFLATE_DBG -I modkit_amd/csrc -o /tmp/inflate_variants tools/dbg/inflate_variants.hip && /tmp/inflate_variants $BAM
/tmp/inflate_variants $BAM 6000
