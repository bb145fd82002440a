#!/bin/bash
# chr1 at real size, phase by phase with walls: generator, `mkpileup pileup` (default and -f 1.0) with --stats, the oracle on all usable CPUs, sha256s
TAG=${1:-chr1}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
P=/tmp/chr1; L=${LEN:-248956422}; N=$(( L * 30 / 9994 ))
( time tools/gen_modbam --out $P --contig chr1:$L --reads $N --seed 60 --style hm --cpg-depleted --mean-len 8353 --threads 16 ) > $OUT/gen.json 2> $OUT/gen.time; tail -3 $OUT/gen.time | head -1; ls -la $P.bam | cut -c1-80
for F in "" "-f 1.0"; do
  T=$(echo $F | tr -d ' -.'); 
  ( time timeout 300 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/chr1_dev$T.bed --cpg --ref $P.fa --stats $F ) > /dev/null 2> $OUT/dev$T.err; echo "device [$F] rc $?"; grep -E "rows=|ahead=|real" $OUT/dev$T.err | cut -c1-330
done
# (the oracle's own -f 1.0 run does not finish in 700 s at this size: only with FULL_ORACLE=1)
for F in "" ${FULL_ORACLE:+"-f 1.0"}; do
  T=$(echo $F | tr -d ' -.');
  ( time timeout ${ORACLE_TIMEOUT:-600} oracle/modkit_oracle pileup $P.bam /tmp/chr1_ora$T.bed --cpg --ref $P.fa --oracle-workers 16 $F ) > /dev/null 2> $OUT/ora$T.err; echo "oracle [$F] rc $?"; grep -E "rows=|real" $OUT/ora$T.err | cut -c1-300
  sha256sum /tmp/chr1_dev$T.bed /tmp/chr1_ora$T.bed | tee -a $OUT/sha.txt
done
sha256sum /tmp/chr1_dev.bed /tmp/chr1_devf10.bed | tee -a $OUT/sha.txt
