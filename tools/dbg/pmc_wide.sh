#!/bin/bash
# wide counter sweep over the bench kernels: what bounds mkp_decode_slots* (issue, LDS, vector memory, scalar memory, instruction fetch)?
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcw}; mkdir -p $OUT; cd /tmp
rocprofv3 -L > $OUT/counters_all.txt 2>&1
grep -oE "\b(SQ|SQC|TCP|TA|TD|TCC|GRBM)_[A-Za-z0-9_]+" $OUT/counters_all.txt | sort -u > $OUT/counter_names.txt; wc -l $OUT/counter_names.txt
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
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass 2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass 3 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_ACTIVE_INST_MISC
pass 4 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_STALL
pass 5 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass 6 TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum
pass 7 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum
pass 8 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
pass 9 GRBM_GUI_ACTIVE GRBM_COUNT SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_INSTS_VALU_MFMA_I8
