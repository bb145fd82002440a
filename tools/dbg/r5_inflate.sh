#!/bin/bash
# round 5: mkp_inflate_wave4 (parallel output step) against round 4's kernels — corpus test, then ms per launch on the C3 bench BAM
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5_inflate_pytest.log
python tools/dbg/inflate_bench.py > /dev/null 2>&1   # generates /tmp/inflate_c3.bam, warms the page cache
for k in wave4 wave4_4k wave4_16k wave3 thread2; do
  for n in 4000 16000 0; do
    if [ $n = 0 ]; then unset INFLATE_BLOCKS; else export INFLATE_BLOCKS=$n; fi
    echo "kernel $k blocks $n: $(MKP_INFLATE_KERNEL=$k python tools/dbg/inflate_bench.py /tmp/inflate_c3.bam 2>&1 | tail -1)"
  done
done > gpurun_out/r5_inflate_sweep.txt 2>&1
unset INFLATE_BLOCKS
cat gpurun_out/r5_inflate_pytest.log gpurun_out/r5_inflate_sweep.txt
