#!/bin/bash
# one-shot run: hipStreamSynchronize returns 10-29 ms after 1 ms of kernels (GPU timestamps) — which runtime knob moves it?
TAG=${1:-r5n}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp MKP_BENCH_DIR=/tmp
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_w.bed --cpg --ref $P.fa > /dev/null 2>&1
for V in "A=1" "ROC_ACTIVE_WAIT_TIMEOUT=100000" "GPU_MAX_HW_QUEUES=8" "AMD_DIRECT_DISPATCH=0" "HIP_FORCE_DEV_KERNARG=1" "ROC_ACTIVE_WAIT_TIMEOUT=100000 GPU_MAX_HW_QUEUES=8"; do
  r=""
  for i in 1 2 3 4 5 6; do rm -f /tmp/o_t.bed; env $V MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_t.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/t.err; r="$r $(grep -o 'total_ms=[0-9.]*' $OUT/t.err | cut -d= -f2)/$(grep 'kernels: sync' $OUT/t.err | awk '{print $(NF-1)}')"; done
  echo "$V: total_ms/sync_ms$r"
done
