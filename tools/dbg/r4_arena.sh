#!/bin/bash
TAG=${1:-r4x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null 2>&1
./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa > /dev/null 2>&1
for A in 0 1 2 0 1 2; do
  MKP_ARENA_PREALLOC=$A MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa --stats 2> $OUT/cli_a$A.txt
  echo "prealloc=$A: $(grep -E 'kernels: sync|run: fetch rows' $OUT/cli_a$A.txt | awk '{print $(NF-1)}' | tr '\n' ' ') ingest-in-hand $(grep 'ingest in hand' $OUT/cli_a$A.txt | awk '{print $(NF-1)}') closed $(grep 'output closed' $OUT/cli_a$A.txt | awk '{print $(NF-1)}')"
done
