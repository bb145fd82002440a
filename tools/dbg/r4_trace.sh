#!/bin/bash
# C3 end to end with the planner / kernel-phase laps
TAG=${1:-r4t}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export MKP_BENCH_DIR=/tmp
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null 2>&1
for k in 1 2 3; do MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa --stats 2> $OUT/cli_$k.txt; done
grep -v "mkpileup plan\] \(dup\|caller\|decode\|prob\|sorted\|slot\|cand\|upload\)" $OUT/cli_3.txt | grep "mkpileup" | cut -c1-200
