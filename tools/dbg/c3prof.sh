cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/gen_modbam --out /tmp/c3 --contig chr20:5400000 --reads 16000 --seed 20 --style hm --cpg-depleted --mean-len 10000 --threads 8 > /dev/null
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3p -o c3 -- $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup /tmp/c3.bam /tmp/c3.bed --cpg --ref /tmp/c3.fa --filter-threshold 0.66 > /dev/null 2>&1
cat $(find /tmp/c3p -name '*kernel_stats.csv') | head -8
grep -E "mkp_emit" $(find /tmp/c3p -name '*kernel_trace.csv') | head -3 | cut -c1-400
head -1 $(find /tmp/c3p -name '*kernel_trace.csv')
