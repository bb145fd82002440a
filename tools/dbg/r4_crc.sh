#!/bin/bash
TAG=${1:-r4y}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_loud_failures.py -m gpu -x -q 2>&1 | tail -6 | cut -c1-300
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null 2>&1
for k in 1 2 3; do MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa --stats 2> $OUT/cli_$k.txt; done
grep -E "mkpileup ingest\]|output closed" $OUT/cli_2.txt $OUT/cli_3.txt | cut -c1-330
sha256sum /tmp/o.bed
