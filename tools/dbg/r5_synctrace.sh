#!/bin/bash
# where do the ~20 ms between the step kernels' launch and hipStreamSynchronize's return go in a one-shot run?  kernel + HIP API timeline of one CLI run
TAG=${1:-r5l}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT; R=$PWD
export TMPDIR=/tmp
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_w.bed --cpg --ref $P.fa > /dev/null 2>&1
cd /tmp; rm -rf /tmp/prof_tl
timeout 300 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -o tl -- $R/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_tl.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli.err; echo "exit $?"
ls -la $(find /tmp/prof_tl -name '*.csv')
python3 - <<'PY' > $OUT/timeline.txt
import csv, glob
def load(pat):
    f = glob.glob('/tmp/prof_tl/**/*' + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = load('kernel_trace.csv'); h = load('hip_api_trace.csv'); m = load('memory_copy_trace.csv')
t0 = min(int(r['Start_Timestamp']) for r in k)
ev = []
for r in k: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].split('(')[0][:40] + ' q' + r.get('Queue_Id', '?')))
for r in m: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
for r in h:
    n = r['Function']
    if n in ('hipStreamSynchronize', 'hipDeviceSynchronize', 'hipEventSynchronize', 'hipMalloc', 'hipFree', 'hipHostMalloc', 'hipHostFree', 'hipMemcpy', 'hipMemset', 'hipHostRegister') and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 300000:
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'H ' + n + ' tid' + r.get('Thread_Id', '?')))
ev.sort()
# the last 60 ms before the final pileup kernel + 40 ms after
kend = max(e for s, e, n in ev if n.startswith('K mkp_pileup_stream'))
for s, e, n in ev:
    if kend - 80e6 < s < kend + 60e6: print('%9.3f %9.3f %8.3f ms  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
tail -80 $OUT/timeline.txt
