#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3i; mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_gpu_extract.py -q ) > $OUT/pytest_extract.log 2>&1; echo "pytest extract exit $?"; tail -n 12 $OUT/pytest_extract.log | cut -c1-700
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --scale 0.125 --dist-backend gloo ) > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "bench n2 exit $?"; tail -5 $OUT/bench_n2.err | cut -c1-600
python - <<PY
import json
try:
    for l in open("$OUT/bench_n2.json"):
        if l.startswith("{"):
            d=json.loads(l); c=d["config"]; print("n2 value %.3g ms/step %.3f scaling %s"%(d["value"], d["ms_per_step"], d["scaling"])); print({k:c.get(k) for k in ("sharded_equals_single_gpu","resident_windows_equal_sharded","imbalance_pileup_wall_max_over_mean")}); print(c.get("per_rank"))
except Exception as e: print("n2 parse failed", e)
PY
