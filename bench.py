#!/usr/bin/env python
"""bench.py — `modkit pileup` hot path on MI355X: genomic positions/s (and bedMethyl rows/s).

A step = one pass of the device pipeline over HBM-resident shards: decode (focus runs: the slot decoder mkp_decode_slots* writing
the per-read feature stream; otherwise the event decoders) -> accumulate + emit (mkp_pileup_stream / mkp_pileup_tiles*) ->
mkp_scan_tiles + mkp_gather_rows.

Workload at N=1 (default): BASELINE.json configs[2] "C3" — synthetic hg38-chr20-sized contig (64 444 167 bp, CpG-depleted
first-order chain), 193 000 reads of mean ~10 kb (~30x), 5mC+5hmC calls alternating `C+hm?` / `C+h?;C+m?` on every read CpG,
`--cpg --ref`, default interval size and default 10th-percentile threshold: the largest single-GPU configuration.
Other workloads (not the headline line): `--workload c2` (configs[1]: 5 Mb contig, 100 000 reads, `C+m?`, all positions), `hemi`
(pileup-hemi on duplex reads), `c4` / `c5` (configs[3] / [4] as scale models of the 24-contig genome: `--genome-scale`, default
1/10 of the hg38 lengths; c4 = 30x `--preset traditional`, c5 = 60x `C+h?;C+m?;A+a?` + per-mod thresholds + `--include-bed`).

N > 1 (`--gpus N`, one process per GPU): ONE BAM of N chr20-sized contigs (C3's generator, N x the reads) is sharded over the ranks
by modkit_amd.distributed.pileup_sharded — contiguous runs of the reference's interval grid balanced by the bytes the BAI puts under
them, every rank reading only its own BGZF blocks, thresholds from the histogram all-reduce over RCCL (the path's one collective,
thresholds.rs:121-159) — and the timed region re-launches every rank's HBM-resident windows.  Per-GPU work is fixed as N grows
("scaling": "weak"), the file and the sharding are north_star's.  The concatenated output is sha256-compared with a single-GPU
run of the same file.

Tiers: kernels only on the resident shard (`value`), the device pipeline (pack + H2D + kernels + D2H), end to end
(`modkit pileup` wall: BGZF inflate, threshold sampling, focus, device pipeline, bedMethyl text) and the per-interval C-ABI seam
(tests/abi_client.c: one mkp_shard_begin/add_records/run per 100 kb interval), next to the CPU restatement of the reference's path
(oracle/, NOT the modkit binary: no Rust toolchain here) timed on the same BAM on this box's host cores, sha256 of both outputs
compared, with the host thread counts of both sides stated and a matched-thread comparison.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import random
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HG38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973),
        ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718),
        ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983),
        ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]
MEAN_ALIGNED = 9994.0   # aligned bases per read of the generator at --mean-len 8353 (median of the log-normal)

WORKLOADS = {
    # name: (style flags of the generator, pileup flags (FASTA / BED substituted), description)
    "c3": (["--style", "hm", "--cpg-depleted", "--mean-len", "8353"], ["--cpg", "--ref", "{fa}"],
           "C3: synthetic hg38 chr20 (CpG-depleted chain), ~30x reads of mean ~10 kb, C+hm? / C+h?;C+m? alternating at every read CpG, --cpg --ref, -i 100000, default 10th-percentile threshold"),
    "hemi": (["--style", "duplex", "--cpg-depleted", "--mean-len", "8353"], ["--cpg", "--ref", "{fa}"],
             "pileup-hemi on the C3 geometry: duplex reads, C+hm?;G-hm? / C+h?;C+m?;G-h?;G-m? alternating at every read CpG, --cpg -r, -i 100000, default 10th-percentile threshold"),
    "c2": (["--style", "m"], [],
           "C2: synthetic 1 contig x 5 Mb, 100 000 reads (~96x), C+m? at every read CpG, all positions, default 10th-percentile threshold"),
    "chr1": (["--style", "hm", "--cpg-depleted", "--mean-len", "8353"], ["--cpg", "--ref", "{fa}"],
             "chr1: ONE contig of hg38 chr1's length x {gs} (a contig longer than a shard: 2^27 positions / 1 GiB of BAM), CpG-depleted chain, 30x, C+hm? / C+h?;C+m?, --cpg --ref, default sharding"),
    "c4": (["--style", "hm", "--cpg-depleted", "--mean-len", "8353"], ["--preset", "traditional", "--ref", "{fa}"],
           "C4 scale model: 24 contigs at {gs} of the hg38 lengths (CpG-depleted chain), 30x, C+hm? / C+h?;C+m?, --preset traditional (= --cpg --combine-strands --ignore h) --ref"),
    "c5": (["--style", "hma", "--cpg-depleted", "--mean-len", "8353"],
           ["--mod-thresholds", "m:0.8", "--mod-thresholds", "h:0.9", "--mod-thresholds", "a:0.7", "--include-bed", "{bed}"],
           "C5 scale model: 24 contigs at {gs} of the hg38 lengths, 60x, C+h?;C+m?;A+a? (6mA at every A), --mod-thresholds m:0.8 h:0.9 a:0.7 over estimated per-base thresholds, --include-bed = seeded random 2 kb intervals (seed 5, mixed BED3 / BED6) at the full genome's density"),
}


def usable_cpus():
    """CPUs this process may use: affinity mask cut by the cgroup CPU quota (the GPU box's container: 256 hardware threads, cpu.max = 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def sh256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def gen_bam(prefix, contigs, n_reads, seed, flags, threads):
    tool = os.path.join(ROOT, "tools", "gen_modbam")
    meta = prefix + ".json"
    if not (os.path.exists(prefix + ".bam") and os.path.exists(prefix + ".bam.bai") and os.path.exists(meta)):
        cmd = [tool, "--out", prefix, "--reads", str(n_reads), "--seed", str(seed), "--threads", str(threads)] + flags
        for name, ln in contigs:
            cmd += ["--contig", "%s:%d" % (name, ln)]
        out = subprocess.check_output(cmd)
        with open(meta, "w") as f:
            f.write(out.decode())
    return prefix + ".bam", prefix + ".fa", json.load(open(meta))


def gen_bed(path, contigs, n, seed=5, width=2000):
    """BASELINE configs[4]: seeded random 2 kb intervals, mixed BED3 / BED6 (+ / - / .)"""
    rng = random.Random(seed)
    total = sum(l for _, l in contigs)
    with open(path, "w") as f:
        for i in range(n):
            x = rng.randrange(total)
            for name, ln in contigs:
                if x < ln:
                    break
                x -= ln
            s = max(0, min(x, ln - width))
            kind = rng.randrange(4)
            f.write("%s\t%d\t%d\n" % (name, s, s + width) if kind == 0 else "%s\t%d\t%d\tiv%d\t0\t%s\n" % (name, s, s + width, i, "+-."[kind - 1]))
    return path


def stages_of(rep):
    """The stages of one mkp_pileup_run as the report gives them, plus what they leave of the wall.  ingest_wait = what the run was blocked
    on the device ingest (BGZF blocks up, inflate, record cut, packing) — before the threshold estimate when it samples from the resident
    shard, in the shard loop otherwise; grid_wait = the estimate blocked on the reference FASTA + interval grid; focus runs on a second
    thread beside the estimate (its time is not on the critical path unless grid_wait says so) and is left out of the sum."""
    st = {"ingest_wait": rep.load_ms, "grid_wait": rep.grid_wait_ms, "threshold": rep.threshold_ms, "threshold_callback": rep.callback_ms, "pack": rep.pack_ms, "h2d": rep.h2d_ms,
          "kernels": rep.kernel_ms, "d2h": rep.d2h_ms, "bedmethyl_text_write": rep.write_ms}
    st["other (plan, launches, syncs, allocation)"] = rep.total_ms - sum(st.values())
    st["focus (beside the estimate)"] = rep.focus_ms
    return st


def ingest_roofline(rep):
    """The device ingest's dominant kernel against the HBM roof: the inflate reads the compressed blocks and writes what they inflate to."""
    if not rep.ingest_kernel_ms or not rep.ingest_raw_bytes:
        return None
    nbytes = rep.ingest_comp_bytes + rep.ingest_raw_bytes
    ach = nbytes / (rep.ingest_kernel_ms * 1e-3) / 1e9
    return {"kernel": "mkp_inflate_wave4 (+ record chains)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": int(nbytes),
            "avg_launch_ms": rep.ingest_kernel_ms / max(1, rep.n_shards), "launches": int(rep.n_shards), "inflated_GBps": rep.ingest_raw_bytes / (rep.ingest_kernel_ms * 1e-3) / 1e9,
            "compressed_bytes": int(rep.ingest_comp_bytes), "inflated_bytes": int(rep.ingest_raw_bytes), "blocks": int(rep.ingest_blocks), "records": int(rep.ingest_records),
            "host_ms": {"uploads": rep.ingest_upload_ms, "block_tables": rep.ingest_table_ms, "parse_scan_pack": rep.ingest_pack_ms},
            "what": "DEFLATE decode is instruction-bound (speculative decode over 64 bit positions per wave), not bandwidth-bound: the fraction of the HBM roof is reported because the contract asks for it; HIP events around the inflate launch(es) of the end-to-end pass, summed over its shards"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(bam, flags, workers, region=None, hemi=False, tag="full"):
    """The oracle (CPU restatement of the reference's path, NOT the reference binary) on the bench BAM itself; `region` bounds the sample."""
    oracle = os.path.join(ROOT, "oracle", "modkit_oracle")
    if not os.path.exists(oracle):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "modkit_oracle"], stdout=subprocess.DEVNULL)
    out = bam + ".oracle.%s.bed" % tag
    rflags = ["--region", region] if region else []
    t0 = time.time()
    p = subprocess.run([oracle] + (["pileup-hemi", bam, "-o", out] if hemi else ["pileup", bam, out]) + ["--oracle-workers", str(workers)] + flags + rflags, capture_output=True, text=True)
    wall = time.time() - t0
    if p.returncode != 0:
        raise RuntimeError("oracle failed: " + p.stderr[-400:])
    m = re.search(r"rows=(\d+) positions=(\d+).*load_s=([0-9.]+) threshold_s=([0-9.]+) pileup_s=([0-9.]+) total_s=([0-9.]+)", p.stderr)
    rows, positions = int(m.group(1)), int(m.group(2))
    load_s, thr_s, pileup_s, total_s = (float(m.group(i)) for i in (3, 4, 5, 6))
    what = "the whole bench workload" if not region else "region %s of the bench BAM (the BAM load and the threshold sample still cover the whole file)" % region
    return out, {
        "value": positions / pileup_s, "unit": "positions/s", "cores": workers, "kind": "port", "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "usable_cpus": usable_cpus(),
        "sample": "%s; restated CPU path (oracle/, interval-parallel like the reference's Rayon pool, %d worker threads); value = pileup phase only with the BAM already decoded in RAM" % (what, workers),
        "rows_per_s": rows / pileup_s, "positions": positions, "rows": rows,
        "end_to_end": {"positions_per_s": positions / total_s, "rows_per_s": rows / total_s, "total_s": total_s, "load_s": load_s, "threshold_s": thr_s, "pileup_s": pileup_s, "wall_s": wall},
    }


def device_e2e_subprocess(bam, out, flags, pool_threads, hemi=False):
    """`modkit pileup` end to end in a fresh process with the library's host pool capped at `pool_threads` (MKP_POOL_THREADS is read
    once per process): wall of mkp_pileup_main as the CLI reports it."""
    cli = os.path.join(ROOT, "modkit_amd", "csrc", "mkpileup")
    env = dict(os.environ, MKP_POOL_THREADS=str(pool_threads), MKP_PACK_PIECES=str(pool_threads))
    t0 = time.time()
    p = subprocess.run([cli, "pileup-hemi" if hemi else "pileup", bam] + (["-o", out] if hemi else [out]) + flags + ["--stats"], capture_output=True, text=True, env=env)
    wall = time.time() - t0
    if p.returncode != 0:
        return {"error": p.stderr[-300:]}
    m = re.search(r"total_ms=([0-9.]+)", p.stderr)
    return {"host_threads": pool_threads, "total_ms": float(m.group(1)) if m else None, "process_wall_s": wall}


def seam_per_interval(bam, fa, out, thr, per_batch=None):
    """The seam a Rust maintainer would call: tests/abi_client.c drives mkp_shard_begin / add_records / run once per
    100 kb interval on the bench BAM; its own stderr line carries intervals, rows and the time inside the API calls."""
    exe = os.path.join(os.environ.get("MKP_BENCH_DIR", "/tmp"), "mkp_abi_client")
    lib_dir = os.path.join(ROOT, "modkit_amd", "csrc")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "abi_client.c"),
                                   "-L", lib_dir, "-lmkpileup", "-lz", "-Wl,-rpath," + lib_dir])
        t0 = time.time()
        p = subprocess.run([exe, bam, fa, out, "cpg", repr(float(thr))] + (["100000", str(per_batch)] if per_batch else []), capture_output=True, text=True, timeout=600)
        wall = time.time() - t0
        if p.returncode != 0:
            return {"error": p.stderr[-300:]}
        kv = dict(re.findall(r"(\w+)=([0-9.eE+-]+)", p.stderr))
        return {"intervals": int(float(kv.get("intervals", 0))), "rows": int(float(kv.get("rows", 0))), "rows_per_s_api": float(kv.get("rows_per_s_api", 0)),
                "api_s": float(kv.get("api_s", 0)), "process_wall_s": wall,
                "what": "tests/abi_client.c: own FASTA reader and focus bytes, one mkp_process_region per contig — the library reads the BAM itself (device ingest: compressed blocks up, records cut and packed in HBM) — with a fixed pass threshold; rates over the time spent inside the API call, which includes reading the file" if per_batch == "file" else
                        ("tests/abi_client.c: own BGZF/BAM/FASTA readers, one mkp_batch_run per %d consecutive 100 kb intervals (process_region_batch's MultiChromCoordinates) with a fixed pass threshold; rates over the time spent inside the API call" % per_batch) if per_batch else
                        "tests/abi_client.c: own BGZF/BAM/FASTA readers, one mkp_shard_begin / mkp_shard_add_records / mkp_shard_run per 100 kb interval with a fixed pass threshold; rates over the time spent inside the three API calls"}
    except Exception as e:  # noqa: BLE001 — the seam tier is informative, never fatal
        return {"error": str(e)[-300:]}


def pmc_traffic(argv_tail, timeout_s=420):
    """HBM bytes per launch of every mkp_* kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950
    note of MI355X_MICROARCH.md, WRITE_SIZE as reported; KiB -> bytes) over a short inner run of this same bench on this build."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mkp_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--inner", "--steps", "2", "--warmup", "1"] + argv_tail
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 --pmc %s timed out" % counter
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 --pmc %s failed: %s" % (counter, p.stderr[-200:])
        tot, n = {}, {}
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0]
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
                n[k] = n.get(k, 0) + 1
        res[counter] = {k: tot[k] / n[k] for k in tot}
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k in res["FETCH_SIZE"]:
        if k.startswith("mkp_"):
            out[k] = int((2.0 * res["FETCH_SIZE"][k] + res["WRITE_SIZE"].get(k, 0.0)) * 1024)
            out[k + ":read"] = int(2.0 * res["FETCH_SIZE"][k] * 1024)
            out[k + ":write"] = int(res["WRITE_SIZE"].get(k, 0.0) * 1024)
    return out, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --inner --steps 2 --warmup 1` of this build, mean per launch, "
                 "bytes = (2*FETCH + WRITE) * 1024; the factor 2 is calibrated on this GPU for streams and byte walks alike (profiles/r02_pmc_calibration.txt)")


def threshold_argv(thr_h):
    t = []
    for i in range(4):
        if thr_h[i] > 0:
            t += ["--filter-threshold", "%s:%r" % ("ACGT"[i], thr_h[i])]
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink a single-contig workload (debug only; the JSON says so)")
    ap.add_argument("--genome-scale", type=float, default=0.1, help="c4 / c5: fraction of the hg38 contig lengths (the JSON states it next to every number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", choices=["full", "region"], default="full", help="CPU baseline + parity on the whole bench BAM (default) or on its first eighth")
    ap.add_argument("--cpu-whole", action="store_true", help="c4 / c5: CPU baseline + sha256 parity on the WHOLE scale model (all usable CPUs; minutes), not on its last contig")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)  # the short run the --pmc passes profile
    ap.add_argument("--tile", type=int, default=0, help="experiments: reference positions per accumulate tile (0 = the library's plan)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl", help="nccl (= RCCL over xGMI, one GPU per rank) or gloo (tests: several ranks may share a GPU)")
    ap.add_argument("--no-full-data", action="store_true", help="c4 / chr1: skip the extra -f 1.0 run (and, with --cpu-whole, the oracle's)")
    ap.add_argument("--skip-e2e", action="store_true", help="profiling runs: skip the sharded end-to-end pass so that every kernel launch is the full-size one the timed region repeats")
    a = ap.parse_args()
    if a.skip_e2e or a.inner:
        a.no_cpu_baseline = True   # no sharded output to compare with

    import torch
    import modkit_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d ranks: launch with python -m torch.distributed.run --nproc-per-node %d ..." % (a.gpus, a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in libmkpileup)")
    if a.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()   # tests: ranks share the GPUs there are
    torch.cuda.set_device(local_rank)
    tdev = "cuda" if a.dist_backend == "nccl" else "cpu"      # where the few scalars that cross ranks live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    gflags, pflags, desc = WORKLOADS[a.workload]
    multi = a.workload in ("c4", "c5", "chr1")
    if a.workload == "chr1" and a.genome_scale == 0.1:
        a.genome_scale = 1.0   # (the point of this workload is the real length; --genome-scale < 1 only for dry runs)
    hemi = a.workload == "hemi"
    if multi and world > 1:
        raise SystemExit("--workload %s is a single-process run (the multi-GPU line shards the C3 generator's BAM)" % a.workload)
    tmp = os.environ.get("MKP_BENCH_DIR", "/tmp")
    # ---- the workload's BAM (rank 0 generates; the generator binary normally ships prebuilt, __graft_entry__.build())
    if multi:
        contigs = [(n, max(100_000, int(l * a.genome_scale))) for n, l in (HG38[:1] if a.workload == "chr1" else HG38)]
        cov = 60 if a.workload == "c5" else 30
        n_reads = int(cov * sum(l for _, l in contigs) / MEAN_ALIGNED)
        seed = {"c4": 40, "c5": 50, "chr1": 60}[a.workload]
        tag = "mkp_%s_g%g" % (a.workload, a.genome_scale)
    else:
        base_len, base_reads = (5_000_000, 100_000) if a.workload == "c2" else (64_444_167, 193_000)
        cl, nr = int(base_len * a.scale), int(base_reads * a.scale)
        name = "synth5m" if a.workload == "c2" else "chr20"
        contigs = [(name, cl)] if world == 1 else [("%s_%d" % (name, k), cl) for k in range(world)]   # N > 1: one BAM, N contigs, sharded over the ranks
        n_reads = nr * world
        seed = {"c3": 20, "hemi": 30}.get(a.workload, 1)
        tag = "mkp_%s_L%d_N%d_x%d" % (a.workload, cl, nr, world)
    prefix = os.path.join(tmp, "%s_seed%d" % (tag, seed))
    t0 = time.time()
    if rank == 0:
        if not os.path.exists(os.path.join(ROOT, "tools", "gen_modbam")):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)
        gen_bam(prefix, contigs, n_reads, seed, gflags, usable_cpus())
        if a.workload == "c5":
            gen_bed(prefix + ".bed", contigs, max(50, int(20000 * a.genome_scale)))
    if dist:
        dist.barrier()
    bam, fa, meta = gen_bam(prefix, contigs, n_reads, seed, gflags, 1)
    gen_s = time.time() - t0
    total_len = sum(l for _, l in contigs)
    # `-t 8`: the reference's --threads (8 here, its default is 4) sets how many intervals are in flight (chunk size floor(1.5 t)) and how
    # the sampling schedule batches its intervals; the device run and the CPU baseline get the same value so that they sample the same reads
    flags = [f.format(fa=fa, bed=prefix + ".bed") for f in pflags] + ["-t", "8"]
    desc = desc.format(gs="%g" % a.genome_scale) + "; %d contig(s), %d bp, %d reads (mean %.0f aligned bp, ~%.1fx)" % (len(contigs), total_len, meta["reads"], meta["aligned_bases"] / max(1, meta["reads"]), meta["aligned_bases"] / total_len)
    host_threads = int(modkit_amd.lib().mkp_host_threads())   # the library's pool: usable CPUs (affinity, cgroup quota) capped at 64, or MKP_POOL_THREADS

    def run_subcommand(ctx, out, extra):   # `modkit pileup` / `modkit pileup-hemi` on a bench context
        return ctx.pileup_hemi_run([bam, "-o", out] + flags + extra) if hemi else ctx.pileup_run([bam, out] + flags + extra)

    out_bed = bam + ".device.bed"
    extra_cfg, tiers = {}, {}
    if multi:
        # ---- multi-shard workloads: K passes of the whole subcommand; device times are the library's HIP-event sums over the shards
        ctx = modkit_amd.Context(device=local_rank, tile_positions=a.tile)
        rep = run_subcommand(ctx, out_bed, [])
        thr_h = [float(rep.threshold[i]) if rep.has_threshold[i] else 0.0 for i in range(4)]
        kms, walls = [], []
        for k in range(a.warmup + a.steps):
            r = run_subcommand(ctx, out_bed + ".rep", threshold_argv(thr_h))
            if k >= a.warmup:
                kms.append(r.kernel_ms); walls.append(r.total_ms)
        if a.steps and sh256(out_bed + ".rep") != sh256(out_bed):
            raise SystemExit("repeat pass with explicit thresholds differs from the sampled run")
        st = ctx.stats()
        ms_per_step = sum(kms) / max(1, len(kms))
        n_rows, total_positions, total_rows = int(rep.n_rows), float(rep.n_positions), float(rep.n_rows)
        value = total_positions / (ms_per_step * 1e-3)
        extra_cfg = {"genome_scale": a.genome_scale, "shards": int(rep.n_shards), "timing": "multi-shard workload: value = positions / device kernel time summed over the shards of one pass (HIP events inside the library, mean over the timed passes), NOT a wall-clock bracket; walls are in tiers.end_to_end",
                     "pass_wall_ms_mean": sum(walls) / max(1, len(walls)), "kernel_ms_last_shard": {"decode": st.decode_kernel_ms, "pileup": st.pileup_kernel_ms, "gather": st.gather_kernel_ms}}
        if a.workload in ("c4", "chr1") and not a.no_full_data:   # the full-data percentile (-f 1.0) next to the default sampled threshold
            rf = run_subcommand(ctx, out_bed + ".f1", ["-f", "1.0"])
            extra_cfg["full_data_threshold_run"] = {"flag": "-f 1.0", "total_ms": rf.total_ms, "threshold_ms": rf.threshold_ms, "thresholds": {"ACGT"[i]: float(rf.threshold[i]) for i in range(4) if rf.has_threshold[i]}, "rows": int(rf.n_rows)}
        rep1, elapsed = rep, ms_per_step * 1e-3 * a.steps
        ctxs = [ctx]
    elif world == 1:
        ctx = modkit_amd.Context(device=local_rank, tile_positions=a.tile)
        rep = rep_warm = None
        if not (a.inner or a.skip_e2e):
            # end to end: the whole subcommand on this context (block reads + inflate, threshold sampling, focus, device pipeline,
            # bedMethyl text), with the driver's default sharding (the next shard's blocks inflate while this one is packed and run)
            rep = run_subcommand(ctx, out_bed, [])
            thr_h = [float(rep.threshold[i]) if rep.has_threshold[i] else 0.0 for i in range(4)]
            # the same call again on the same context: staging, window and row buffers exist now (what a long-lived caller pays per file)
            rep_warm = run_subcommand(ctx, out_bed + ".warm", [])
            if sh256(out_bed) != sh256(out_bed + ".warm"):
                raise SystemExit("second end-to-end pass on the same context differs from the first")
            os.remove(out_bed + ".warm")
        else:
            thr = ctx.estimate_thresholds(bam, ["-t", "8"])
            thr_h = [float(thr.get(b, 0.0)) for b in "ACGT"]
        # kernels-only tier: the whole contig as ONE HBM-resident shard (same thresholds), re-launched K times
        rep1 = run_subcommand(ctx, out_bed + ".oneshard", threshold_argv(thr_h) + ["--shard-bytes", str(1 << 40)])
        if rep is None:
            rep = rep1
        elif sh256(out_bed) != sh256(out_bed + ".oneshard"):
            raise SystemExit("sharded and single-shard bedMethyl differ")
        # the rows the timed re-launches must reproduce: the one-shard output (sha256-equal to the end-to-end file, which the CPU baseline leg compares with the oracle's)
        expected_row_digests = None if hemi else [modkit_amd.rows_digest(modkit_amd.read_bedmethyl(out_bed + ".oneshard"))]
        if rep is rep1:
            os.replace(out_bed + ".oneshard", out_bed)
        else:
            os.remove(out_bed + ".oneshard")
        ctxs = [ctx]
    else:
        # ---- N ranks, ONE BAM: thresholds from the all-reduced histograms, every rank runs its contiguous run of the interval grid
        from modkit_amd import distributed as mkd
        shard_stats = {}
        # one shard per contig piece a rank owns, as the single-GPU line's one resident shard: the default cuts 8 pieces per rank (balance
        # for small files) and 256 MiB of BAM per shard (host memory), which only adds launch and tile-edge overhead to a resident re-run
        flags = flags + ["--shard-bp", str(1 << 27), "--shard-bytes", str(1 << 40)]
        # the END-TO-END pass of the N-GPU job, timed like the step (barrier + synchronize either side, max over ranks): every rank ingests its
        # shards on its GPU, samples them from HBM (-f 1.0), the histograms are all-reduced (the path's one collective), every rank runs its
        # pileup pass on the resident shards, rank 0 concatenates.  (The all-reduce mode is BASELINE configs[3]'s; the broadcast mode is
        # covered by tests/test_gpu_scale.py.)
        dist.barrier(); torch.cuda.synchronize()
        t_sh = time.perf_counter()
        thr = mkd.pileup_sharded([bam, out_bed] + flags, rank=rank, world=world, device=local_rank, stats=shard_stats, mode="full")
        torch.cuda.synchronize(); dist.barrier()
        sharded_wall_s = time.perf_counter() - t_sh
        thr_h = [float(thr.get(b, 0.0)) for b in "ACGT"]
        # the rank's windows, each resident in HBM on its own context (a window = a piece of one contig; cuts sit on the interval grid)
        plan = mkd.shard_plan([bam, out_bed] + flags, rank, world)
        ctxs, rep1, n_rows_rank, parts, expected_row_digests = [], None, 0, [], []
        for k, (cname, s, e) in enumerate(plan):
            c = modkit_amd.Context(device=local_rank, tile_positions=a.tile)
            part = "%s.rank%d.win%d" % (out_bed, rank, k)
            r = run_subcommand(c, part, threshold_argv(thr_h) + ["--region", "%s:%d-%d" % (cname, s, e), "--shard-bytes", str(1 << 40)])
            n_rows_rank += int(r.n_rows); parts.append(part); ctxs.append(c)
            expected_row_digests.append(modkit_amd.rows_digest(modkit_amd.read_bedmethyl(part)))
            rep1 = r if rep1 is None else rep1
        rep = rep1
        # the windows' rows = the rank's part of the sharded run (validated below through the concatenation)
        with open("%s.rank%d.windows" % (out_bed, rank), "wb") as f:
            for p_ in parts:
                with open(p_, "rb") as g:
                    f.write(g.read())
                os.remove(p_)
        n_rows = n_rows_rank

    if not multi:
        for c in ctxs:
            c.rerun(a.warmup)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if len(ctxs) == 1:
            ctxs[0].rerun(a.steps)  # K passes; each launch sequence ends with a stream sync inside the library
        else:
            for _ in range(a.steps):
                for c in ctxs:
                    c.rerun(1)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        st = ctxs[0].stats()
        # the rows the LAST timed re-launch left in HBM, fetched now (rerun(0): no further launch) and compared with the checked output: the
        # kernels the timed region ran are the kernels whose rows are bit-exact, on this very pass
        timed_rows_checked = None
        if os.environ.get("MKP_DEBUG_SKIP", "0") not in ("", "0"):
            timed_rows_checked = {"equal": None, "what": "skipped: MKP_DEBUG_SKIP ablation run of a -DMKP_DEBUG build (the kernels leave parts out on purpose)"}
        elif expected_row_digests is not None and a.steps > 0:
            for c, want in zip(ctxs, expected_row_digests):
                got = modkit_amd.rows_digest(modkit_amd.rows_to_numpy(c.rerun(0, fetch=True)))
                if got != want:
                    raise SystemExit("rows of the last timed re-launch differ from the checked bedMethyl output (digest %s vs %s)" % (got, want))
            timed_rows_checked = {"equal": True, "contexts": len(ctxs), "sha256_of_row_columns": expected_row_digests[0],
                                  "what": "mkp_shard_rerun(0, fetch): the row columns left by the last timed launch == the columns of the bedMethyl file this run checked (sha256 over pos, strand, code, n_valid .. n_nocall)"}
        if a.inner:
            for c in ctxs:
                c.close()
            return
        if world == 1:
            n_rows = int(rep.n_rows)
        my_positions = float(total_len) if world == 1 else float(sum(e - s for _, s, e in plan))
        el = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        if dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        totals = torch.tensor([my_positions, float(n_rows)], dtype=torch.float64, device=tdev)
        if dist:
            dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        total_positions, total_rows = float(totals[0].item()), float(totals[1].item())
        ms_per_step = elapsed / a.steps * 1e3
        value = total_positions * a.steps / elapsed
        if world > 1:
            # per-rank figures (gathered to rank 0): windows, positions, reads, BAM bytes under the rank's run, kernel ms per step, walls
            mine = {"rank": rank, "windows": len(plan), "positions": my_positions, "rows": n_rows, "reads_first_window": int(st.n_reads), "ms_per_step": elapsed / a.steps * 1e3,
                    "threshold_s": shard_stats.get("threshold_s"), "pileup_s": shard_stats.get("pileup_s"), "total_s": shard_stats.get("total_s"), "part_bytes": shard_stats.get("part_bytes"),
                    "report": shard_stats.get("report"), "sharded_wall_s": sharded_wall_s}
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            if rank == 0:
                # parity of the split: concatenation of the ranks' parts == a single-GPU run of the same file with the same thresholds;
                # and the per-window resident runs reproduce the parts
                single = out_bed + ".single"
                modkit_amd.pileup([bam, single, "--device", str(local_rank)] + flags + threshold_argv(thr_h))
                win_cat = hashlib.sha256()
                for r_ in range(world):
                    with open("%s.rank%d.windows" % (out_bed, r_), "rb") as g:
                        win_cat.update(g.read())
                walls = [g_["pileup_s"] for g_ in gathered if g_["pileup_s"]]
                e2e_wall = max(g_["sharded_wall_s"] for g_ in gathered)
                tiers["end_to_end_sharded"] = {"positions_per_s": total_positions / e2e_wall, "rows_per_s": total_rows / e2e_wall, "ms": e2e_wall * 1e3,
                                               "per_rank_total_s": [g_["total_s"] for g_ in gathered], "per_rank_threshold_s": [g_["threshold_s"] for g_ in gathered],
                                               "what": "the N-GPU job end to end, max over ranks between two barriers: per rank ONE mkp_pileup_run_cb (device ingest of its shards ahead, full-data sample from HBM, histogram all-reduce in the threshold callback, pileup pass on the resident shards, bedMethyl text) + the concatenation on rank 0; page cache warm"}
                extra_cfg = {"sharding": "ONE BAM (%d contigs), contiguous runs of the interval grid per rank balanced by BAI bytes (mkp_pileup_main --gpus-rank/--gpus-world via modkit_amd.distributed.pileup_sharded), thresholds: two-level histogram all-reduce over RCCL (-f 1.0)" % len(contigs),
                             "per_rank": gathered, "imbalance_pileup_wall_max_over_mean": (max(walls) / (sum(walls) / len(walls))) if walls else None,
                             "sharded_sha256": sh256(out_bed), "sharded_equals_single_gpu": sh256(out_bed) == sh256(single), "resident_windows_equal_sharded": win_cat.hexdigest() == sh256(out_bed)}
                os.remove(single)
            dist.barrier()

    if rank == 0:
        focus_run = bool(st.slot_pipeline)
        agg_kernel = "mkp_pileup_tiles_hemi" if hemi else "mkp_pileup_stream" if focus_run else "mkp_pileup_tiles"
        kernels = {"decode": (st.decode_kernel_ms, st.alg_bytes_decode), "aggregate": (st.pileup_kernel_ms, st.alg_bytes_pileup)}
        # SURVEY §8(d) one-pass algorithmic bytes of the whole pass: reads (16 + 4 n_cigar + L/2) + calls (2 + K) + 44 per row
        if focus_run:
            b_alg = st.alg_bytes_decode - 5 * st.stream_bytes - 32 * st.n_reads + 44 * st.n_rows
        else:
            b_alg = st.alg_bytes_decode - 8 * st.n_events + 44 * st.n_rows
        traffic_all, traffic_src = None, None
        if world == 1 and not a.no_pmc and not multi:
            tail = ["--workload", a.workload, "--scale", str(a.scale), "--no-cpu-baseline", "--no-pmc"]
            traffic_all, traffic_src = pmc_traffic(tail)
        def kernel_traffic(prefixes):
            if not traffic_all:
                return None
            return sum(v for k, v in traffic_all.items() if ":" not in k and any(k.startswith(p) for p in prefixes)) or None
        dec_prefixes = ("mkp_decode_", "mkp_cover_reads", "mkp_merge_duplex", "mkp_hemi_failed")
        def roof(name, ms, nbytes, traffic):
            ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": ms, "traffic": traffic}
        agg = roof(agg_kernel, st.pileup_kernel_ms, st.alg_bytes_pileup, (traffic_all or {}).get(agg_kernel))
        agg.update({"traffic_read": (traffic_all or {}).get(agg_kernel + ":read"), "traffic_write": (traffic_all or {}).get(agg_kernel + ":write"), "traffic_source": traffic_src,
                    "accounting": ("feature stream (1 B per read and focus position) + 32 B visit record per read + 44 B per row" if focus_run else "reads (16 + 4 n_cigar + L/2) + 8 B per call event + 44 B per row")})
        if focus_run:   # SURVEY §8(d)'s literal B_agg: 8 B per coverage event at a candidate position + 44 B per row
            s_ach = st.alg_bytes_agg_survey / (st.pileup_kernel_ms * 1e-3) / 1e9
            agg["survey_B_agg"] = {"bytes": int(st.alg_bytes_agg_survey), "achieved": s_ach, "frac": s_ach / HBM_PEAK_GBS, "what": "SURVEY §8(d): 8 B x (read, candidate position) coverage events + 44 B x rows over the same launch time; the kernel itself reads the events as 1-byte features"}
        slowest_name = "decode" if st.decode_kernel_ms >= st.pileup_kernel_ms else "aggregate"
        slow = roof("mkp_decode_slots* (+ mkp_cover_reads)" if focus_run else "mkp_decode_*", st.decode_kernel_ms, st.alg_bytes_decode, kernel_traffic(dec_prefixes)) if slowest_name == "decode" else dict(agg)
        whole = None
        if not multi:
            whole = roof("whole pass: decode + aggregate + gather", ms_per_step, b_alg, None)
            whole["what"] = "SURVEY §8(d) one-pass B_alg (reads + calls + 44 B rows, every byte counted once) over the driver-timed step"
        dev_ms = rep1.pack_ms + rep1.h2d_ms + rep1.kernel_ms + rep1.d2h_ms
        tiers.update({
            "kernels_only": {"positions_per_s": value, "rows_per_s": total_rows * a.steps / elapsed if elapsed else 0.0, "ms": ms_per_step, "what": "timed region: K re-launches on the HBM-resident shard(s)"},
            "device_pipeline": {"positions_per_s": rep1.n_positions / (dev_ms * 1e-3), "rows_per_s": rep1.n_rows / (dev_ms * 1e-3), "ms": dev_ms,
                                "stages_ms": {"pack": rep1.pack_ms, "h2d": rep1.h2d_ms, "kernels": rep1.kernel_ms, "d2h": rep1.d2h_ms},
                                "what": "rank 0: planner + uploads of the plan + kernels + D2H of rows for a pass whose reads the device ingest has already packed in HBM (the BAM side of the pass is tiers.end_to_end's ingest_wait)" + ("" if multi else ", one shard")},
            "end_to_end": {"positions_per_s": rep.n_positions / (rep.total_ms * 1e-3), "rows_per_s": rep.n_rows / (rep.total_ms * 1e-3), "ms": rep.total_ms, "host_threads": host_threads,
                           "stages_ms": stages_of(rep),
                           "shards": int(rep.n_shards), "what": "rank 0: mkp_pileup_run wall (`modkit pileup in.bam out.bed` with the workload's flags, default sharding), page cache warm, the library's host pool at host_threads threads; bam_load_inflate = what the shard loop waited for blocks (the rest overlaps with pack / run / write)"},
        })
        if world == 1 and not multi and locals().get("rep_warm") is not None:
            rw = rep_warm
            tiers["end_to_end_warm_context"] = {"positions_per_s": rw.n_positions / (rw.total_ms * 1e-3), "rows_per_s": rw.n_rows / (rw.total_ms * 1e-3), "ms": rw.total_ms,
                                                "stages_ms": stages_of(rw),
                                                "what": "the same mkp_pileup_run again on the same context (page-locked staging, ingest windows and row buffers already allocated); tiers.end_to_end is the first call on a fresh context"}
        if world == 1 and a.workload == "c3" and not (a.inner or a.skip_e2e):
            tiers["seam_per_interval"] = seam_per_interval(bam, fa, bam + ".seam.bed", thr_h[1] if thr_h[1] > 0 else 0.7)
            # the batch seam (mkp_batch_run): the reference's default chunk of floor(1.5 * threads) intervals at 8 and 64 threads, and a whole contig per call
            tiers["seam_batch"] = {str(n): seam_per_interval(bam, fa, bam + ".seam.bed", thr_h[1] if thr_h[1] > 0 else 0.7, per_batch=n) for n in (12, 96, 1000)}
            # the file seam (mkp_process_region): the caller hands over the path, not the records
            tiers["seam_file"] = seam_per_interval(bam, fa, bam + ".seam.bed", thr_h[1] if thr_h[1] > 0 else 0.7, per_batch="file")
        result = {
            "metric": "genomic positions/sec pileup (bedMethyl rows/s); bit-exact vs ref", "value": value, "unit": "positions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            # `value` re-launches the kernels on shards that are resident in HBM (the contract's "inputs already resident"); this is the same
            # job from the BAM file on disk to the bedMethyl file on disk (tiers.end_to_end / tiers.end_to_end_sharded), in the same unit
            "value_end_to_end": (tiers["end_to_end_sharded"]["positions_per_s"] if "end_to_end_sharded" in tiers else tiers["end_to_end"]["positions_per_s"]),
            "config": dict({"workload": desc + ("; ONE BAM of %d such contigs sharded over %d ranks" % (world, world) if world > 1 else ""),
                            "scale": a.scale, "rows_per_s": total_rows * a.steps / elapsed if elapsed else 0.0, "rows_per_step": total_rows, "reads": int(st.n_reads), "call_events": int(st.n_events),
                            "tiles": int(st.n_tiles), "thresholds": {"ACGT"[i]: thr_h[i] for i in range(4) if thr_h[i] > 0},
                            "pipeline": "slot pipeline (feature stream)" if focus_run else "event pipeline (tile walk)",
                            "kernel_ms": {"decode": st.decode_kernel_ms, "pileup": st.pileup_kernel_ms, "rows": st.rows_kernel_ms, "gather": st.gather_kernel_ms},
                            "generator_s": gen_s, "arithmetic": "u32 tallies in LDS; f32 threshold caller (bit-exact vs the reference's f32)",
                            "parity": "bit-exact vs the restated CPU path (oracle/) on this BAM; the oracle is pinned on the reference's golden files; ties / >=3 codes / QC-fail / N ops are reference-unpinned and sample-probs has no reference pin (DESIGN.md §7)",
                            "slowest_kernel": slowest_name, "timed_rows_checked": locals().get("timed_rows_checked"),
                            "ingest": "device (compressed BGZF blocks up; inflate, record cut, MM/ML tokeniser and packing in HBM)" if not os.environ.get("MKP_HOST_INGEST") == "1" else "host",
                            "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MKP_") and k != "MKP_BENCH_DIR"}}, **extra_cfg),
            "tiers": tiers,
            # `roofline` itself = the aggregation kernel (north_star's target); `dominant` = the kernel that takes most of the step
            "roofline": dict(agg, slowest=slow, dominant=dict(slow, share_of_step=(slow["avg_launch_ms"] / ms_per_step) if ms_per_step else None), whole_pass=whole, ingest=ingest_roofline(rep)),
        }
        if world == 1 and not a.no_cpu_baseline:
            workers = min(usable_cpus(), 8)
            region = None
            if multi and a.cpu_whole:
                workers = usable_cpus()   # the whole scale model: every contig's rows compared (sha256), the oracle on all usable CPUs
            elif multi:
                region = "%s:0-%d" % (contigs[-1][0], contigs[-1][1])   # bounded sample: the last (shortest-but-one) contig
            elif a.cpu_sample == "region":
                region = "%s:0-%d" % (contigs[0][0], contigs[0][1] // 8)
            obed, base = cpu_baseline(bam, flags, workers, region, hemi, "s" if region else "full")
            if region:
                dbed = bam + ".device.region.bed"
                (modkit_amd.pileup_hemi if hemi else modkit_amd.pileup)(([bam, "-o", dbed] if hemi else [bam, dbed]) + ["--device", str(local_rank)] + flags + ["--region", region])
            else:
                dbed = out_bed
            base["bedmethyl_sha256_equal"] = sh256(dbed) == sh256(obed)
            base["bedmethyl_sha256"] = sh256(dbed)
            if a.workload == "chr1" and os.path.exists(out_bed + ".f1"):
                # (the oracle's -f 1.0 run reads and sorts the whole contig's probabilities on one thread: > 700 s at this size, profiles/r06_chr1.txt)
                base["full_data_threshold_run"] = {"flag": "-f 1.0", "equals_default_threshold_output": sh256(out_bed + ".f1") == sh256(out_bed)}
            elif multi and a.cpu_whole and os.path.exists(out_bed + ".f1"):
                # the full-data estimate (-f 1.0: sampled from the shards resident in HBM) against the oracle's -f 1.0 run, whole output
                obed1, base1 = cpu_baseline(bam, flags + ["-f", "1.0"], workers, None, hemi, "full_f1")
                base["full_data_threshold_run"] = {"flag": "-f 1.0", "bedmethyl_sha256_equal": sh256(out_bed + ".f1") == sh256(obed1), "oracle_total_s": base1["end_to_end"]["total_s"], "oracle_threshold_s": base1["end_to_end"]["threshold_s"]}
            dev_pps = rep.n_positions / (rep.total_ms * 1e-3)
            base["speedup_end_to_end"] = dev_pps / base["end_to_end"]["positions_per_s"] if not region else None
            base["device_host_threads"] = host_threads
            if not region and not hemi and not multi:
                # matched host thread counts: the device run's host side capped at 8 threads against the oracle on 8; both at all usable CPUs (at most 64)
                m8 = device_e2e_subprocess(bam, bam + ".device.t8.bed", flags, 8)
                matched = {"8_threads": {"device": m8, "oracle_total_s": base["end_to_end"]["total_s"],
                                         "speedup_end_to_end": (base["end_to_end"]["total_s"] * 1e3 / m8["total_ms"]) if m8.get("total_ms") else None}}
                if usable_cpus() > 8:
                    w2 = min(usable_cpus(), 64)
                    _, b2 = cpu_baseline(bam, [f for f in flags if f not in ("-t", "8")] + ["-t", str(w2)], w2, None, hemi, "t%d" % w2)   # (-t steers its sampling schedule too: timing only, no sha comparison)
                    mN = device_e2e_subprocess(bam, bam + ".device.t%d.bed" % w2, flags, w2)
                    matched["%d_threads" % w2] = {"device": mN, "oracle_total_s": b2["end_to_end"]["total_s"], "oracle_positions_per_s": b2["value"],
                                                  "speedup_end_to_end": (b2["end_to_end"]["total_s"] * 1e3 / mN["total_ms"]) if mN.get("total_ms") else None}
                base["matched_host_threads"] = matched
            result["cpu_baseline"] = base
        print(json.dumps(result))
    for c in ctxs:
        c.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
