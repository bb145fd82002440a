#!/bin/bash
# rocprofv3 kernel stats of the two device inflate kernels on the bench BAM (whole file in one launch) and of --device-inflate end to end
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/inflate_prof; mkdir -p $OUT; export TMPDIR=/tmp
for K in wave thread; do MKP_INFLATE_KERNEL=$K timeout 300 python tools/dbg/inflate_bench.py > $OUT/inflate_$K.json 2> /dev/null; cut -c1-300 $OUT/inflate_$K.json; done
cd /tmp; rm -rf /tmp/pi; P=/tmp/inflate_c3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -o p -- $GRAFT_REPO_ROOT/modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_dev.bed --cpg --ref $P.fa --stats --device-inflate > /dev/null 2> $OUT/e2e_device.err
for f in $(find /tmp/pi -name '*kernel_stats.csv'); do cp $f $OUT/device_inflate_e2e_kernel_stats.csv; done
head -6 $OUT/device_inflate_e2e_kernel_stats.csv | cut -c1-150
