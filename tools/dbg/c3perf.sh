set -e
cd $GRAFT_REPO_ROOT
make -C tools gen_modbam >/dev/null
tools/gen_modbam --out /tmp/c3 --contig chr20:5400000 --reads 16000 --seed 20 --style hm --cpg-depleted --mean-len 10000 --threads 8
for i in 1 2; do modkit_amd/csrc/mkpileup pileup /tmp/c3.bam /tmp/c3.bed --cpg --ref /tmp/c3.fa --stats --rerun 10; done
tools/gen_modbam --out /tmp/c5 --contig chr1:3000000 --reads 9000 --seed 5 --style hma --mean-len 10000 --threads 8
for i in 1 2; do modkit_amd/csrc/mkpileup pileup /tmp/c5.bam /tmp/c5.bed --filter-threshold 0.7 --stats --rerun 10; done
wc -l /tmp/c3.bed /tmp/c5.bed
