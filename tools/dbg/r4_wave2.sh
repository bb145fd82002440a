#!/bin/bash
# the speculative wave inflate kernel: corpus tests of all four kernels, launch-size sweep against thread2 / wave, C3 and C4 end to end with it;
# and what RCCL says when ncclCommInitRank fails on one rank
TAG=${1:-r4j}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q 2>&1 | tail -25 | cut -c1-600 > $OUT/pytest_inflate.log; cat $OUT/pytest_inflate.log
for K in wave2 thread2 wave; do for NB in 0 16000 4000 1000; do
  if [ $NB = 0 ]; then unset INFLATE_BLOCKS; else export INFLATE_BLOCKS=$NB; fi
  echo -n "$K $NB: "; MKP_INFLATE_KERNEL=$K timeout 300 python tools/dbg/inflate_bench.py 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['inflated_bytes'], [round(r['kernel_ms'],2) for r in d['runs']], round(d['kernel_GBps_inflated'],1), 'GB/s')"
done; done 2>&1 | tee $OUT/sweep.txt
unset INFLATE_BLOCKS
export MKP_BENCH_DIR=/tmp
for K in wave2 default; do
  if [ $K = default ]; then unset MKP_INFLATE_KERNEL; else export MKP_INFLATE_KERNEL=$K; fi
  MKP_TRACE_PLAN=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline > $OUT/c3_$K.json 2> $OUT/c3_$K.err
  grep "mkpileup ingest" $OUT/c3_$K.err | tail -2 | cut -c1-330
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/c3_$K.json") if l.startswith("{")][-1])
for k in ("end_to_end","end_to_end_warm_context"):
    print("$K", k, round(d["tiers"][k]["ms"],1), {a:round(b,1) for a,b in d["tiers"][k]["stages_ms"].items()}, d["tiers"]["end_to_end"].get("bedmethyl_sha256_equal"))
PY
done
export MKP_INFLATE_KERNEL=wave2
timeout 600 python bench.py --workload c4 --steps 1 --warmup 0 --no-pmc --no-cpu-baseline > $OUT/c4_wave2.json 2> /dev/null
python - <<PY
import json
d=json.loads([l for l in open("$OUT/c4_wave2.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
print("c4 wave2 e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()})
PY
unset MKP_INFLATE_KERNEL
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k rccl -s 2>&1 | grep -v "^$" | tail -40 | cut -c1-260 > $OUT/rccl_debug.log
MKP_RCCL_LIB=/opt/rocm/lib/librccl.so.1 timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k rccl 2>&1 | tail -3 > $OUT/rccl_optrocm.log
timeout 120 python - > $OUT/torch_nccl.log 2>&1 <<PY
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
dist.init_process_group("nccl", rank=0, world_size=1)
t=torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize(); print("torch nccl one-rank all_reduce ok", t.sum().item())
PY
tail -3 $OUT/torch_nccl.log; tail -12 $OUT/rccl_debug.log; cat $OUT/rccl_optrocm.log
