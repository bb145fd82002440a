#!/bin/bash
# SQ counter pass over the bench kernels (issue-bound vs latency-bound diagnosis)
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-sq}; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/sq -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc ${BENCH_ARGS:-} > /dev/null 2> $OUT/sq.err
f=$(find /tmp/sq -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY' | tee $OUT/sq.txt
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not k.startswith("mkp_"): continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in tot:
    print(k, {c: round(v) for c, v in tot[k].items()})
PY
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/sq2 -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc ${BENCH_ARGS:-} > /dev/null 2> $OUT/sq2.err
f=$(find /tmp/sq2 -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY' | tee $OUT/sq2.txt
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not k.startswith("mkp_"): continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in tot:
    print(k, {c: round(v) for c, v in tot[k].items()})
PY
tail -3 $OUT/sq.err $OUT/sq2.err | cut -c1-300
