#!/bin/bash
# round 4: quick GPU loop — inflate kernel variants on the C3 BAM + the ingest tests + one traced end-to-end run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r4q}
exec > gpurun_out/$TAG.log 2>&1
set -x
export MKP_BENCH_DIR=/tmp
python - <<'PY'
import bench, os, sys
sys.argv=['bench.py']
PY
BAM=/tmp/mkp_c3_L64444167_N193000_x1_seed20.bam; FA=${BAM%.bam}.fa
if [ ! -f $BAM ]; then ./tools/gen_modbam --out ${BAM%.bam} --reads 193000 --seed 20 --threads 16 --style hm --cpg-depleted --mean-len 8353 --contig chr20:64444167 > ${BAM%.bam}.json; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMKP_INFLATE_DBG -I modkit_amd/csrc -o /tmp/inflate_variants tools/dbg/inflate_variants.hip && /tmp/inflate_variants $BAM
timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -25
for k in thread2 thread2; do
  MKP_INFLATE_KERNEL=$k MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $BAM /tmp/o_$k.bed --cpg --ref $FA -t 8 --stats 2> gpurun_out/${TAG}_trace_$k.txt
  grep -E "mkpileup run|device ingest|total_ms|mkpileup ingest|threshold sampling" gpurun_out/${TAG}_trace_$k.txt
done
MKP_HOST_INGEST=1 ./modkit_amd/csrc/mkpileup pileup $BAM /tmp/o_host.bed --cpg --ref $FA -t 8 --stats 2> gpurun_out/${TAG}_trace_host.txt
cmp /tmp/o_thread2.bed /tmp/o_host.bed && echo thread2_EQUALS_HOST
