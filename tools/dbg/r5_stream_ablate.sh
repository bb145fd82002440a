#!/bin/bash
# round 5: (1) the staged ingest (upload || block table || inflate) — parity + the C3 timeline; (2) where mkp_pileup_stream's time goes:
# MKP_DEBUG_SKIP on the debug build — 1024 no visits, 2048 no scans / rows, 4096 no look-back (rows in completion order)
TAG=${1:-r5f}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity_golden.py tests/test_gpu_inflate.py tests/test_gpu_parity_hemi.py -x -q -m gpu ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
P=/tmp/r5_c3
[ -f $P.bam ] || tools/gen_modbam --out $P --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 16 > /dev/null
modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_warm.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_warm.err
for i in 1 2 3; do MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_cli2.bed --cpg --ref $P.fa --stats > /dev/null 2> $OUT/cli_trace$i.err; grep -E "ingest\]|total_ms" $OUT/cli_trace$i.err | cut -c1-420 | head -4; done
sha256sum /tmp/o_cli2.bed
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench_c3.json 2> $OUT/bench_c3.err; python - <<PY
import json
d=json.load(open("$OUT/bench_c3.json"))
print("c3", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"], "e2e_ms %.0f"%d["tiers"]["end_to_end"]["ms"], {k: round(v) for k, v in d["tiers"]["end_to_end"]["stages_ms"].items()})
print("   warm", {k: round(v) for k, v in d["tiers"]["end_to_end_warm_context"]["stages_ms"].items()}, round(d["tiers"]["end_to_end_warm_context"]["ms"]))
print("   ingest", json.dumps(d["roofline"]["ingest"])[:420])
PY
export MKP_LIB_PATH=$PWD/tools/dbg/lib/libmkpileup_debug.so
for K in 0 1024 2048 3072 4096; do
  MKP_DEBUG_SKIP=$K timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc --no-cpu-baseline > $OUT/ab_$K.json 2> $OUT/ab_$K.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$K.json")); print("skip $K", "ms/step %.3f"%d["ms_per_step"], d["config"]["kernel_ms"])
except Exception as e: print("skip $K failed", open("$OUT/ab_$K.err").read()[-300:])
PY
done
MKP_PILEUP_WAVES=4 MKP_DEBUG_SKIP=0 timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc --no-cpu-baseline > $OUT/ab_w4.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/ab_w4.json')); print('w4', d['config']['kernel_ms'])"
for K in 1024 2048; do MKP_PILEUP_WAVES=4 MKP_DEBUG_SKIP=$K timeout 300 python bench.py --steps 20 --warmup 3 --skip-e2e --no-pmc --no-cpu-baseline > $OUT/ab_w4_$K.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/ab_w4_$K.json')); print('w4 skip $K', d['config']['kernel_ms'])"; done
