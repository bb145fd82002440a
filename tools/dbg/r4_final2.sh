#!/bin/bash
# after the one-shot / scratch-free accumulate build: the whole suite, the default bench line, the c4 line, a CLI trace, the c3 kernel stats
TAG=${1:-r04b}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
export MKP_BENCH_DIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/c3_bench.json 2> $OUT/c3_bench.err; echo "bench c3 exit $?"
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
for k in 1 2; do MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa --stats 2> $OUT/cli_$k.txt; done
grep -E "kernels: sync|run: kernels|run: fetch|total_ms|output closed|ingest in hand" $OUT/cli_2.txt | cut -c1-260
timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $OUT/c4_bench.json 2> /dev/null
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $OUT/c5_bench.json 2> /dev/null
cd /tmp; rm -rf /tmp/prof_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --skip-e2e > /dev/null 2> $OUT/c3_rocprof.err
for f in $(find /tmp/prof_c3 -name '*kernel_stats.csv'); do cp $f $OUT/c3_kernel_stats.csv; done
python - <<PY
import json
d=json.loads([l for l in open("$OUT/c3_bench.json") if l.startswith("{")][-1]); t=d["tiers"]
print("c3 ms/step", d["ms_per_step"], d["config"]["kernel_ms"], "sha", d["cpu_baseline"].get("bedmethyl_sha256_equal"))
for k in ("end_to_end","end_to_end_warm_context"): print(k, round(t[k]["ms"],1), {a:round(b,1) for a,b in t[k]["stages_ms"].items()})
print("seam", t["seam_per_interval"].get("rows_per_s_api"), {k:v.get("rows_per_s_api") for k,v in t["seam_batch"].items()}, t["seam_file"].get("rows_per_s_api"), t["seam_file"].get("api_s"))
for w in ("c4","c5"):
    e=json.loads([l for l in open("$OUT/%s_bench.json"%w) if l.startswith("{")][-1])["tiers"]["end_to_end"]
    print(w, "e2e ms %.0f"%e["ms"], {k:round(v) for k,v in e["stages_ms"].items()})
PY
