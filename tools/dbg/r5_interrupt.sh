#!/bin/bash
# the 15-28 ms between the step kernels' launch and hipStreamSynchronize's return in a one-shot run: interrupt wake-up?  HSA_ENABLE_INTERRUPT=0 (polling signals)
TAG=${1:-r5m}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp MKP_BENCH_DIR=/tmp
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_w.bed --cpg --ref $P.fa > /dev/null 2>&1
for V in "A=1" "HSA_ENABLE_INTERRUPT=0" "A=1" "HSA_ENABLE_INTERRUPT=0"; do
  r=""
  for i in 1 2 3 4; do rm -f /tmp/o_t.bed; env $V MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_t.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/t.err; r="$r $(grep -o 'total_ms=[0-9.]*' $OUT/t.err | cut -d= -f2) (sync $(grep 'kernels: sync' $OUT/t.err | awk '{print $(NF-1)}'), ingest $(grep -o 'total [0-9.]* ms, of which' $OUT/t.err | awk '{print $2}'))"; done
  echo "$V: total_ms$r"
done
for V in "A=1" "HSA_ENABLE_INTERRUPT=0"; do
  env $V timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("$V bench c3", "ms/step %.3f"%d["ms_per_step"], "e2e_ms %.0f"%d["tiers"]["end_to_end"]["ms"], {k: round(v) for k, v in d["tiers"]["end_to_end"]["stages_ms"].items()})
print("   warm", round(d["tiers"]["end_to_end_warm_context"]["ms"]), {k: round(v) for k, v in d["tiers"]["end_to_end_warm_context"]["stages_ms"].items()})
PY
done
