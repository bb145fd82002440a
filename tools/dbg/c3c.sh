cd $GRAFT_REPO_ROOT
tools/gen_modbam --out /tmp/c3 --contig chr20:5400000 --reads 16000 --seed 20 --style hm --cpg-depleted --mean-len 10000 --threads 8 >/dev/null
for S in 0 256 512; do echo "skip=$S"; MKP_DEBUG_SKIP=$S modkit_amd/csrc/mkpileup pileup /tmp/c3.bam /tmp/c3.bed --cpg --ref /tmp/c3.fa --filter-threshold 0.66 --stats --rerun 10 2>&1 | grep "rows=" | sed -E 's/.*rows=([0-9]+).*kernel_ms=([0-9.]+ \([^)]*\)).*/rows=\1 \2/'; done
