#!/bin/bash
# wide counter sweep over the bench kernels: what bounds mkp_decode_slots* (issue, LDS, vector memory, scalar memory, instruction fetch)?
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcw}; mkdir -p $OUT; cd /tmp
[ -n "$LIST" ] && rocprofv3 -L > $OUT/counters_all.txt 2>&1
[ -n "$LIST" ] && grep -oE "\b(SQ|SQC|TCP|TA|TD|TCC|GRBM)_[A-Za-z0-9_]+" $OUT/counters_all.txt | sort -u > $OUT/counter_names.txt; wc -l $OUT/counter_names.txt
pass() { n=$1; shift
  rm -rf /tmp/pw$n
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pw$n -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --skip-e2e ${BENCH_ARGS:-} > /dev/null 2> $OUT/p$n.err
  f=$(find /tmp/pw$n -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $n failed: $(tail -2 $OUT/p$n.err | cut -c1-200)"; return; fi
  python - "$f" <<'PY' | tee -a $OUT/pmc.txt
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not (k.startswith("mkp_decode") or k.startswith("mkp_pileup") or k.startswith("mkp_cover")): continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k, {c: round(v / max(1, cnt[k][c])) for c, v in sorted(tot[k].items())})
PY
}
PASSES=${PASSES:-1 2}
for P in $PASSES; do case $P in
1) pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU;;
2) pass 2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS;;
3) pass 3 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_ACTIVE_INST_MISC;;
4) pass 4 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_STALL;;
# (TCP_* / TA_* / TCC_* multi-counter passes hang rocprofv3 on this pool: 300 s each, never again)
esac; done
