#!/bin/bash
# wave-per-block inflate: parity tests, throughput against the one-thread-per-block kernel, and --device-inflate end to end
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/device_inflate; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q ) > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
for K in wave thread; do MKP_INFLATE_KERNEL=$K timeout 300 python tools/dbg/inflate_bench.py > $OUT/inflate_$K.json 2> $OUT/inflate_$K.err; echo "$K: $(cut -c1-400 $OUT/inflate_$K.json)"; tail -2 $OUT/inflate_$K.err | cut -c1-300; done
if [ -n "$E2E" ]; then
P=/tmp/inflate_c3
for M in host device device; do F=""; [ $M = device ] && F="--device-inflate"
  MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_$M.bed --cpg --ref $P.fa --stats $F 2> $OUT/trace_$M.txt; echo "$M: $(grep -o 'load_ms=[0-9.]*\|threshold_ms=[0-9.]*\|pack_ms=[0-9.]*\|total_ms=[0-9.]*\|on the device [0-9]*' $OUT/trace_$M.txt | tr '\n' ' ')"; done
sha256sum /tmp/o_host.bed /tmp/o_device.bed
fi
