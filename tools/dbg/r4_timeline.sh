#!/bin/bash
# GPU timeline (kernel dispatches + memory copies with timestamps) of one `mkpileup pileup` run on the C3 bench BAM
TAG=${1:-r4u}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
[ -f $P.bam ] || $GRAFT_REPO_ROOT/tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null 2>&1
$GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa > /dev/null 2>&1
rm -rf /tmp/tl; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o tl -- $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli.err
for f in $(find /tmp/tl -name '*kernel_trace.csv'); do cp $f $OUT/kernel_trace.csv; done
for f in $(find /tmp/tl -name '*memory_copy_trace.csv'); do cp $f $OUT/memory_copy_trace.csv; done
ls -la $OUT; head -2 $OUT/kernel_trace.csv | cut -c1-400; head -2 $OUT/memory_copy_trace.csv | cut -c1-400
