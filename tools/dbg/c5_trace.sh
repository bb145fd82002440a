#!/bin/bash
# the C5 scale model through the CLI with the planner / ingest / sampler traces on
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; TAG=${1:-c5t}
export MKP_BENCH_DIR=/tmp
python - <<'PY'
import sys, os
sys.argv = ["bench.py"]
sys.path.insert(0, ".")
import bench, random
contigs = [(n, max(100_000, int(l * 0.1))) for n, l in bench.HG38]
n_reads = int(60 * sum(l for _, l in contigs) / bench.MEAN_ALIGNED)
prefix = "/tmp/mkp_c5_g0.1_seed50"
bench.gen_bam(prefix, contigs, n_reads, 50, bench.WORKLOADS["c5"][0], bench.usable_cpus())
bench.gen_bed(prefix + ".bed", contigs, max(50, int(20000 * 0.1)))
PY
P=/tmp/mkp_c5_g0.1_seed50
for k in 1 2; do
MKP_TRACE_PLAN=1 ./modkit_amd/csrc/mkpileup pileup $P.bam /tmp/c5.bed --mod-thresholds m:0.8 --mod-thresholds h:0.9 --mod-thresholds a:0.7 --include-bed $P.bed -t 8 --stats 2> gpurun_out/${TAG}_$k.txt
done
grep -E "mkpileup ingest|threshold sampling|total_ms|device ingest|thresholds done|shard plan done" gpurun_out/${TAG}_2.txt | cut -c1-330 | head -60
