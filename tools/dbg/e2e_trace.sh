#!/bin/bash
# timeline of `mkpileup pileup` on the C3 bench BAM (MKP_TRACE_PLAN laps + --stats), three runs; the BAM is generated if the bench has not left one
TAG=${1:-e2e}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
P=/tmp/e2e_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_warm.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_warm.err
for i in 1 2 3; do MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli.bed --cpg --ref $P.fa --stats ${EXTRA:-} > /dev/null 2> $OUT/cli_trace$i.err; grep -E "total_ms|threshold sampling" $OUT/cli_trace$i.err | cut -c1-420 | head -3; done
