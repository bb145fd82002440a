#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/pmc_calib.hip).  Usage (on the GPU box): tools/pmc_calib.sh <out-dir>
set -u
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/calib}; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib $GRAFT_REPO_ROOT/tools/pmc_calib.hip || exit 1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$C
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/calib_$C -o calib -- /tmp/pmc_calib > $OUT/run_$C.log 2>&1
  f=$(find /tmp/calib_$C -name '*counter_collection.csv' | head -1)
  python3 - "$f" $C <<'PY' | tee $OUT/calib_$C.txt
import csv, sys, collections
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].split("(")[0]; tot[k] += float(r["Counter_Value"]); n[k] += 1
GiB = 1 << 20   # KiB in 1 GiB
for k in sorted(tot):
    print("%s %s mean_KiB_per_launch %.0f  = %.3f x 1 GiB" % (sys.argv[2], k, tot[k] / n[k], tot[k] / n[k] / GiB))
PY
done
