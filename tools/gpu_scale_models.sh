#!/bin/bash
# BASELINE configs[3] / [4] scale models (bench.py --workload c4 | c5, --genome-scale 0.1) with a bounded CPU baseline (the oracle on the
# last contig of the same BAM), and the 2-rank form of the bench on one GPU (gloo).
# Usage: tools/gpu_scale_models.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-r04s}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export MKP_BENCH_DIR=/tmp
for W in c5 c4; do
  ( time timeout 1500 python bench.py --workload $W --steps 2 --warmup 1 --no-pmc ) > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit $?"; tail -3 $OUT/${W}_bench.err | cut -c1-300
done
( time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo ) > $OUT/n2_gloo_bench.json 2> $OUT/n2_gloo_bench.err; echo "n2 exit $?"
python - <<PY
import json
for w in ("n2_gloo","c4","c5"):
    try:
        d=json.loads([l for l in open("$OUT/%s_bench.json"%w) if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
        cb=d.get("cpu_baseline") or {}
        print(w, "n_gpus", d["n_gpus"], "ms/step %.3f"%d["ms_per_step"], "value %.3g"%d["value"], "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "shards", e.get("shards"), "gen_s", round(d["config"].get("generator_s",0)),
              "cpu", {k: cb.get(k) for k in ("value","cores","sample","bedmethyl_sha256_equal")}, json.dumps(d["config"].get("sharded_equals_single_gpu")))
    except Exception as ex: print(w, "parse failed", ex)
PY
