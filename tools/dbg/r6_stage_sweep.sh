#!/bin/bash
# ingest timeline of `mkpileup pileup` on the C3 BAM under a list of environment settings (MKP_STAGE_ROUNDS=n caps a stage, MKP_STAGE_DEPTH=n stages in flight)
TAG=${1:-stg}; shift; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
for E in "$@"; do
  D=$(echo "$E" | tr ' =' '__')
  env $E bash tools/dbg/e2e_trace.sh $TAG/$D > /dev/null 2>&1
  for i in 1 2 3; do echo "$E: $(grep 'mkpileup ingest' $OUT/$D/cli_trace$i.err | sed 's/.*plan /plan /' | cut -c1-140) $(grep 'mkpileup ingest' $OUT/$D/cli_trace$i.err | grep -o '{inflate.*' | cut -c1-300)"; done
done | tee $OUT/sweep.txt
