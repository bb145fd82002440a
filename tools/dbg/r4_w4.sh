#!/bin/bash
# accumulate kernels with 64 VGPRs + spills (two workgroups per CU) against 128 VGPRs, no scratch (one per CU): step times (c3, c2) and the
# wall of the kernel phase of one CLI run
TAG=${1:-r4w}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export MKP_BENCH_DIR=/tmp
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
for V in 8 4; do
  export MKP_PILEUP_WAVES=$V
  for W in c3 c2; do
    timeout 600 python bench.py --workload $W --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --skip-e2e > $OUT/${W}_w$V.json 2> /dev/null
    python - <<PY
import json
d=json.loads([l for l in open("$OUT/${W}_w$V.json") if l.startswith("{")][-1])
print("$W waves=$V ms/step %.4f"%d["ms_per_step"], d["config"]["kernel_ms"], "sha", d.get("cpu_baseline"))
PY
  done
  for k in 1 2; do MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_$V.bed --cpg --ref $P.fa --stats 2> $OUT/cli_w${V}_$k.txt; done
  grep -E "kernels: sync|run: kernels|run: fetch|total_ms" $OUT/cli_w${V}_2.txt | cut -c1-220
done
cmp /tmp/o_8.bed /tmp/o_4.bed && echo "bedMethyl identical"
