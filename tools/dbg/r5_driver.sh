#!/bin/bash
# round 5: the driver refactor (shards ahead from the plan, HBM budget, full-data sampling from resident shards, mkp_pileup_run_cb) —
# parity of the paths it touches, then the C4 scale model (default mode + -f 1.0) and the 2-rank form of the bench on one GPU (gloo)
TAG=${1:-r5d}; cd "$(dirname "$0")/../.." && OUT=$PWD/gpurun_out/$TAG && mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} MKP_BENCH_DIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity_hemi.py tests/test_gpu_ingest.py -x -q -m gpu -k "not full_size" ) > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
( time timeout 900 python bench.py --workload c4 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline ) > $OUT/c4_bench.json 2> $OUT/c4_bench.err; echo "bench c4 exit $?"; tail -3 $OUT/c4_bench.err | cut -c1-300
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo ) > $OUT/n2_gloo_bench.json 2> $OUT/n2_gloo_bench.err; echo "n2 exit $?"; tail -5 $OUT/n2_gloo_bench.err | cut -c1-300
python - <<PY
import json
for w in ("n2_gloo","c4"):
    try:
        d=json.loads([l for l in open("$OUT/%s_bench.json"%w) if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
        print(w, "n_gpus", d["n_gpus"], "ms/step %.3f"%d["ms_per_step"], "value %.3g"%d["value"], "value_e2e %.3g"%d["value_end_to_end"], "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "shards", e.get("shards"))
        print("   ", json.dumps(d["config"].get("full_data_threshold_run")), json.dumps(d["tiers"].get("end_to_end_sharded")), json.dumps(d["config"].get("sharded_equals_single_gpu")), json.dumps(d["roofline"].get("ingest"))[:600])
    except Exception as ex: print(w, "parse failed", ex)
PY
# the budget: C4 scale model with 1 GiB of HBM for the shards ahead against the default
P=$(ls /tmp/mkp_c4_g0.1_seed40.bam 2>/dev/null); F=${P%.bam}.fa
if [ -n "$P" ]; then
  for B in 0 1024; do
    X=""; [ $B != 0 ] && X="--hbm-budget-mb $B"
    for i in 1 2; do modkit_amd/csrc/mkpileup pileup $P /tmp/o_c4_$B.bed --preset traditional --ref $F -t 8 --stats $X 2> $OUT/c4_cli_budget$B.err > /dev/null; done
    echo "budget $B: $(grep -o 'total_ms=[0-9.]*' $OUT/c4_cli_budget$B.err) $(grep -o 'resident_sampling=.*budget [0-9]* MB)' $OUT/c4_cli_budget$B.err)"
  done
  cmp /tmp/o_c4_0.bed /tmp/o_c4_1024.bed && echo "budget outputs equal"
  modkit_amd/csrc/mkpileup pileup $P /tmp/o_c4_f1.bed --preset traditional --ref $F -t 8 --stats -f 1.0 2> $OUT/c4_cli_f1.err > /dev/null; grep -E "total_ms|threshold sampling|full-data" $OUT/c4_cli_f1.err | cut -c1-300
fi
