#!/bin/bash
# SQ counters of the inflate kernels (16 000 blocks of the bench BAM per launch)
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmci}; mkdir -p $OUT; cd /tmp
export INFLATE_BLOCKS=16000
python $GRAFT_REPO_ROOT/tools/dbg/inflate_bench.py > /dev/null 2>&1   # (writes the BAM)
for K in ${KERNELS:-wave4}; do
  export MKP_INFLATE_KERNEL=$K
  for P in 1 2 3; do
    case $P in
      1) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";;
      2) C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES";;
      3) C="SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC";;
    esac
    rm -rf /tmp/sqi; rocprofv3 --pmc $C --output-format csv -d /tmp/sqi -o sq -- python $GRAFT_REPO_ROOT/tools/dbg/inflate_bench.py > /dev/null 2> $OUT/err_${K}_$P.txt
    f=$(find /tmp/sqi -name '*counter_collection.csv' | head -1)
    python - "$f" $K <<'PY' | tee -a $OUT/sq_inflate.txt
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("mkp_inflate"): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in tot: print(sys.argv[2], k, "(3 launches)", {c: round(v) for c, v in tot[k].items()})
except Exception as e: print(sys.argv[2], "no counters:", e)
PY
  done
done
