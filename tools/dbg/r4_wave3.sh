#!/bin/bash
# mkp_inflate_wave3 (8 KiB ring, ten waves per CU, far matches from the flushed output): corpus test, launch-size sweep against wave2, c4 with it
TAG=${1:-r4z}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "all_device_kernels" 2>&1 | tail -4 | cut -c1-500
for K in wave3 wave2; do for NB in 0 16000 4000; do
  if [ $NB = 0 ]; then unset INFLATE_BLOCKS; else export INFLATE_BLOCKS=$NB; fi
  echo -n "$K $NB: "; MKP_INFLATE_KERNEL=$K timeout 200 python tools/dbg/inflate_bench.py 2>&1 | tail -1 | python -c "import json,sys; l=sys.stdin.read(); d=json.loads(l) if l.startswith('{') else None; print([round(r['kernel_ms'],2) for r in d['runs']], round(d['kernel_GBps_inflated'],1), 'GB/s') if d else print('FAILED', l[-300:])"
done; done 2>&1 | tee $OUT/sweep.txt
unset INFLATE_BLOCKS
export MKP_BENCH_DIR=/tmp MKP_INFLATE_KERNEL=wave3
( timeout 400 python bench.py --workload c4 --steps 1 --warmup 0 --no-pmc ) > $OUT/c4_wave3.json 2> $OUT/c4_wave3.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/c4_wave3.json") if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
    print("c4 wave3 e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()}, "sha equal:", (d.get("cpu_baseline") or {}).get("bedmethyl_sha256_equal"))
except Exception as ex: print("c4 failed", ex, open("$OUT/c4_wave3.err").read()[-400:])
PY
