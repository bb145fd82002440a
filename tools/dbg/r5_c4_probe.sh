#!/bin/bash
# why do the sampling rounds of the C4 scale model take 6 ms each beside the ingest (1.2 ms alone)?  DMA queueing (HSA_ENABLE_SDMA=0), number of
# concurrent ingests (MKP_AHEAD_WORKERS), no ingest beside them at all (MKP_NO_RESIDENT_SAMPLING keeps the host sampler: different path)
TAG=${1:-r5i}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp MKP_BENCH_DIR=/tmp
P=/tmp/mkp_c4_g0.1_seed40; [ -f $P.bam ] || python bench.py --workload c4 --steps 1 --warmup 0 --no-pmc --no-cpu-baseline > /dev/null 2>&1
run() { for i in 1 2; do env "$@" modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_c4.bed --preset traditional --ref $P.fa -t 8 --stats 2> $OUT/err.txt > /dev/null; done
  echo "$* :: $(grep -o 'total_ms=[0-9.]*' $OUT/err.txt) $(grep -o 'load_ms=[0-9.]* threshold_ms=[0-9.]*' $OUT/err.txt) | $(grep -o 'head fetch wait.*' $OUT/err.txt | cut -c1-120)"; }
run A=1
run HSA_ENABLE_SDMA=0
run MKP_AHEAD_WORKERS=1
run MKP_AHEAD_WORKERS=2
run MKP_AHEAD_WORKERS=8
run MKP_HOST_BLOCK_TABLE=1
