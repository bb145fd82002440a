#!/bin/bash
# where the ahead-ingest of the c4 scale model spends its time: per-shard trace with allocation and kernel times
TAG=${1:-r4l}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export MKP_BENCH_DIR=/tmp
for K in default wave2; do
  if [ $K = default ]; then unset MKP_INFLATE_KERNEL; else export MKP_INFLATE_KERNEL=$K; fi
  MKP_TRACE_PLAN=1 timeout 600 python bench.py --workload c4 --steps 1 --warmup 0 --no-pmc --no-cpu-baseline > $OUT/c4_$K.json 2> $OUT/c4_$K.err
  grep -c "mkpileup ingest" $OUT/c4_$K.err
done
