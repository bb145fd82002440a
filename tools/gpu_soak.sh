#!/bin/bash
# Soak run on the GPU box: the fuzzed parity suites (device == oracle, byte for byte) under seeds the suite does not pin.
# Usage: tools/gpu_soak.sh <tag> <first shift> <last shift>     -> gpurun_out/<tag>/soak.txt
TAG=${1:-soak}; A=${2:-1}; B=${3:-4}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
: > $OUT/soak.txt
for S in $(seq $A $B); do
  MKP_SOAK_SHIFT=$S MKP_FUZZ_SEEDS=$((S * 1000)):$((S * 1000 + 20)) MKP_FUZZ_PROFILE_SEEDS=$((S * 1000 + 500)):$((S * 1000 + 504)) \
    timeout 600 python -m pytest -q -n 8 -m gpu tests/test_gpu_parity_fuzz.py tests/test_gpu_parity_hemi.py tests/test_gpu_extract.py tests/test_gpu_ingest.py \
      tests/test_gpu_dup_names.py tests/test_gpu_summary.py -k "fuzz or duplicates or device_ingest" > $OUT/soak_$S.log 2>&1
  echo "shift $S: $(tail -1 $OUT/soak_$S.log)" | tee -a $OUT/soak.txt
  grep -E "^FAILED|^ERROR" $OUT/soak_$S.log | head -20 | tee -a $OUT/soak.txt
done
