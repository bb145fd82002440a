cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r03d; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_gpu.log | head -2
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/c3_bench.json 2> $OUT/c3_bench.err; echo "bench c3 exit $?"
python - <<PY
import json
for w in ("c3",):
    d=json.loads([l for l in open("$OUT/%s_bench.json"%w) if l.startswith("{")][-1]); e=d["tiers"]["end_to_end"]
    print(w, "ms/step %.3f"%d["ms_per_step"], "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()})
    c=d.get("cpu_baseline")
    if c: print("  sha", c["bedmethyl_sha256_equal"], "speedup", round(c["speedup_end_to_end"],1), {k:(v["device"]["total_ms"], v["oracle_total_s"], round(v["speedup_end_to_end"],1)) for k,v in c["matched_host_threads"].items()})
PY
