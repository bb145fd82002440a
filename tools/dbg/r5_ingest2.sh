#!/bin/bash
# round 5: device block table + buffered tokeniser + stream priorities — parity (device ingest == host ingest == oracle), then one profiled CLI run on the C3 BAM
TAG=${1:-r5c}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity_golden.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest.log; cat $OUT/pytest.log
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_warm.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_warm.err
for i in 1 2 3; do MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli2.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_trace$i.err; grep -E "ingest\]|total_ms" $OUT/cli_trace$i.err | cut -c1-420 | head -4; done
MKP_HOST_BLOCK_TABLE=1 MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli3.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_trace_hosttable.err; grep -E "ingest\]|total_ms" $OUT/cli_trace_hosttable.err | cut -c1-420 | head -4
cd /tmp; rm -rf /tmp/prof_cli
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli -o cli -- $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/ingest_cli.err; echo "rocprof cli exit $?"
for f in $(find /tmp/prof_cli -name '*kernel_stats.csv'); do cp $f $OUT/ingest_kernel_stats.csv; done
head -22 $OUT/ingest_kernel_stats.csv | cut -c1-160
cmp /tmp/o_cli.bed /tmp/o_warm.bed && cmp /tmp/o_cli3.bed /tmp/o_warm.bed && sha256sum /tmp/o_cli.bed
