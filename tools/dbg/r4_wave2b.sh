#!/bin/bash
# wave2 with the output window: corpus test, launch-size sweep, c4
TAG=${1:-r4m}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "all_device_kernels" 2>&1 | tail -5 | cut -c1-600 > $OUT/pytest_inflate.log; cat $OUT/pytest_inflate.log
for K in wave2; do for NB in 0 16000 4000 1000; do
  if [ $NB = 0 ]; then unset INFLATE_BLOCKS; else export INFLATE_BLOCKS=$NB; fi
  echo -n "$K $NB: "; MKP_INFLATE_KERNEL=$K timeout 300 python tools/dbg/inflate_bench.py 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['inflated_bytes'], [round(r['kernel_ms'],2) for r in d['runs']], round(d['kernel_GBps_inflated'],1), 'GB/s')"
done; done 2>&1 | tee $OUT/sweep.txt
unset INFLATE_BLOCKS
export MKP_BENCH_DIR=/tmp MKP_INFLATE_KERNEL=wave2
timeout 600 python bench.py --workload c4 --steps 1 --warmup 0 --no-pmc --no-cpu-baseline > $OUT/c4_wave2.json 2> /dev/null
python - <<PY
import json
d=json.loads([l for l in open("$OUT/c4_wave2.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
print("c4 wave2 e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()})
PY
