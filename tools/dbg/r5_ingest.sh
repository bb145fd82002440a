#!/bin/bash
# round 5: ingest kernels after the rewrite (inflate wave4 + ring variants, CRC slicing-by-4, coalesced scans, pack split) —
# parity first (inflate corpus, device ingest == host ingest == oracle), then ms per launch, SQ counters, one profiled CLI run on the C3 BAM
TAG=${1:-r5b}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest.log; cat $OUT/pytest.log
python tools/dbg/inflate_bench.py > /dev/null 2>&1   # generates /tmp/inflate_c3.bam, warms the page cache
for k in wave4 wave4_8k wave4_2k; do
  for n in 4000 16000 0; do
    if [ $n = 0 ]; then unset INFLATE_BLOCKS; else export INFLATE_BLOCKS=$n; fi
    echo "kernel $k blocks $n: $(MKP_INFLATE_KERNEL=$k python tools/dbg/inflate_bench.py /tmp/inflate_c3.bam 2>&1 | tail -1 | cut -c1-420)"
  done
done > $OUT/inflate_sweep.txt 2>&1
unset INFLATE_BLOCKS; cat $OUT/inflate_sweep.txt | cut -c1-300
KERNELS="wave4" bash tools/dbg/pmc_inflate.sh $TAG/sqi > /dev/null 2>&1; cat $OUT/sqi/sq_inflate.txt | cut -c1-600
# one whole run of the subcommand on a C3 BAM: the ingest kernels next to the step kernels
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_warm.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_warm.err
cd /tmp; rm -rf /tmp/prof_cli
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli -o cli -- $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/ingest_cli.err; echo "rocprof cli exit $?"
for f in $(find /tmp/prof_cli -name '*kernel_stats.csv'); do cp $f $OUT/ingest_kernel_stats.csv; done
head -16 $OUT/ingest_kernel_stats.csv | cut -c1-160; grep -E "ingest|total_ms|MKP_" $OUT/ingest_cli.err | cut -c1-300 | head
MKP_TRACE_PLAN=1 $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli2.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_trace.err; grep -E "ingest\]|total" $OUT/cli_trace.err | cut -c1-400 | head -8
cmp /tmp/o_cli.bed /tmp/o_warm.bed && sha256sum /tmp/o_cli.bed
