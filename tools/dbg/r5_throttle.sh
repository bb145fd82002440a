#!/bin/bash
# is the 20 ms "kernels: sync" of a one-shot run (and other unexplained waits) the cgroup's CPU throttle?  cpu.stat around C3 runs at several pool sizes
TAG=${1:-r5k}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp MKP_BENCH_DIR=/tmp
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
cat /sys/fs/cgroup/cpu.max; modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_w.bed --cpg --ref $P.fa > /dev/null 2>&1
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for T in 16 14 12 10 8 16; do
  a=$(stat)
  r=""
  for i in 1 2 3; do rm -f /tmp/o_t.bed; MKP_POOL_THREADS=$T MKP_PACK_PIECES=$T MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_t.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/t$T.err; r="$r $(grep -o 'total_ms=[0-9.]*' $OUT/t$T.err | cut -d= -f2) (sync $(grep 'kernels: sync' $OUT/t$T.err | awk '{print $(NF-1)}'), ingest $(grep -o 'total [0-9.]* ms, of which' $OUT/t$T.err | awk '{print $2}'))"; done
  b=$(stat)
  echo "pool $T: total_ms$r"; echo "   before: $a"; echo "   after:  $b"
done
