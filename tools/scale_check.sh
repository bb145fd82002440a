#!/bin/bash
# The multi-GPU bench line at 1, 2, 4 and 8 GPUs of one node, as the driver launches it, with the parity flags of every line checked:
# the concatenated sharded output must equal the single-GPU run of the same file with the same thresholds.  (Efficiency is the
# driver's to compute from the per-N values.)
# Usage: tools/scale_check.sh [steps] [warmup]     (needs as many visible GPUs as the largest N it runs; smaller boxes run what fits)
set -u
cd "$(dirname "$0")/.." || exit 1
STEPS=${1:-20}; WARMUP=${2:-3}; PORT=${MASTER_PORT:-29577}
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
rc=0
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "skip N=$N (only $NGPU GPU(s) visible)"; continue; fi
  OUT=/tmp/mkp_scale_N$N.json
  if [ "$N" -eq 1 ]; then python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" --no-pmc --no-cpu-baseline > "$OUT" || rc=1
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" > "$OUT" || rc=1; fi
  python - "$OUT" "$N" <<'PY' || rc=1
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); n = int(sys.argv[2])
c = d["config"]
e2e = d["tiers"].get("end_to_end_sharded") or d["tiers"].get("end_to_end") or {}
print("N=%d value %.4g %s ms/step %.3f | end to end %.4g %s in %.0f ms" % (d["n_gpus"], d["value"], d["unit"], d["ms_per_step"], d.get("value_end_to_end", 0.0), d["unit"], e2e.get("ms", 0.0)),
      "sharded_equals_single_gpu:", c.get("sharded_equals_single_gpu"), "imbalance:", c.get("imbalance_pileup_wall_max_over_mean"), "per-rank total s:", e2e.get("per_rank_total_s"))
assert d["n_gpus"] == n
if n > 1:
    assert c.get("sharded_equals_single_gpu") is True and c.get("resident_windows_equal_sharded") is True, "sharded output differs from the single-GPU run"
PY
done
exit $rc
