#!/usr/bin/env python
"""bench.py — `modkit pileup` hot path on MI355X: genomic positions/s (and bedMethyl rows/s).

A step = one pass of the device pipeline (decode kernels -> mkp_pileup_tiles -> row emission) over one HBM-resident shard.
Workload at N=1 (default): BASELINE.json configs[2] "C3" — synthetic hg38-chr20-sized contig (64 444 167 bp, CpG-depleted
first-order chain), 193 000 reads of mean ~10 kb (~30x), 5mC+5hmC calls alternating `C+hm?` / `C+h?;C+m?` on every read CpG,
`--cpg --ref`, default interval size and default 10th-percentile threshold: the largest single-GPU configuration.
`--workload c2` selects configs[1] (5 Mb contig, 100 000 reads, `C+m?`, no motif).  At N>1 every rank runs one such shard of its
own contig (weak scaling; disjoint contigs need no data-path collective) and the per-base pass thresholds come from a histogram
all-reduce over RCCL (the one collective of the path, thresholds.rs:121-159 over all ranks' sampled probabilities).

Three tiers are reported: kernels only on the resident shard (`value`), the device pipeline
(pack + H2D + kernels + D2H) and end to end (`modkit pileup` wall: BGZF inflate, threshold sampling, focus, device pipeline,
bedMethyl text), next to the CPU restatement of the reference's path (oracle/, NOT the modkit binary: no Rust toolchain here)
timed on the same BAM on this box's host cores, with the sha256 of both bedMethyl outputs compared.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

WORKLOADS = {
    # name: (contig, length, reads, generator flags, pileup flags needing the FASTA, description)
    "c3": ("chr20", 64_444_167, 193_000, ["--style", "hm", "--cpg-depleted", "--mean-len", "8353"], True,
           "C3: synthetic hg38 chr20 (%d bp, CpG-depleted chain), %d reads (mean %.0f bp, ~%.0fx), C+hm? / C+h?;C+m? alternating at every read CpG, --cpg --ref, -i 100000, default 10th-percentile threshold"),
    "hemi": ("chr20", 64_444_167, 193_000, ["--style", "duplex", "--cpg-depleted", "--mean-len", "8353"], True,
             "pileup-hemi on the C3 geometry: synthetic hg38 chr20 (%d bp, CpG-depleted chain), %d duplex reads (mean %.0f bp, ~%.0fx), C+hm?;G-hm? / C+h?;C+m?;G-h?;G-m? alternating at every read CpG, --cpg -r, -i 100000, default 10th-percentile threshold"),
    "c2": ("synth5m", 5_000_000, 100_000, ["--style", "m"], False,
           "C2: synthetic 1 contig x %d bp, %d reads (mean %.0f bp, ~%.0fx), C+m? at every read CpG, default 10th-percentile threshold"),
}


def sh256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def gen_bam(prefix, contig, contig_len, n_reads, seed, flags, threads):
    tool = os.path.join(ROOT, "tools", "gen_modbam")
    meta = prefix + ".json"
    if not (os.path.exists(prefix + ".bam") and os.path.exists(prefix + ".bam.bai") and os.path.exists(meta)):
        out = subprocess.check_output([tool, "--out", prefix, "--contig", "%s:%d" % (contig, contig_len), "--reads", str(n_reads), "--seed", str(seed), "--threads", str(threads)] + flags)
        with open(meta, "w") as f:
            f.write(out.decode())
    return prefix + ".bam", prefix + ".fa", json.load(open(meta))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(bam, flags, contig, contig_len, workers, mode, hemi=False):
    """The oracle (CPU restatement of the reference's path, NOT the reference binary) on the bench BAM itself:
    mode 'full' = the whole workload; 'region' = the first eighth of the contig (bounded sample)."""
    oracle = os.path.join(ROOT, "oracle", "modkit_oracle")
    if not os.path.exists(oracle):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "modkit_oracle"], stdout=subprocess.DEVNULL)
    out = bam + ".oracle.%s.bed" % mode
    region = [] if mode == "full" else ["--region", "%s:0-%d" % (contig, contig_len // 8)]
    t0 = time.time()
    p = subprocess.run([oracle] + (["pileup-hemi", bam, "-o", out] if hemi else ["pileup", bam, out]) + ["--oracle-workers", str(workers)] + flags + region, capture_output=True, text=True)
    wall = time.time() - t0
    if p.returncode != 0:
        raise RuntimeError("oracle failed: " + p.stderr[-400:])
    m = re.search(r"rows=(\d+) positions=(\d+).*load_s=([0-9.]+) threshold_s=([0-9.]+) pileup_s=([0-9.]+) total_s=([0-9.]+)", p.stderr)
    rows, positions = int(m.group(1)), int(m.group(2))
    load_s, thr_s, pileup_s, total_s = (float(m.group(i)) for i in (3, 4, 5, 6))
    what = "the whole bench workload" if mode == "full" else "positions [0, %d) of the bench BAM (--region; the BAM load and the threshold sample still cover the whole file)" % (contig_len // 8)
    return out, region, {
        "value": positions / pileup_s, "unit": "positions/s", "cores": workers, "kind": "port", "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
        "sample": "%s; restated CPU path (oracle/, interval-parallel like the reference's Rayon pool, %d worker threads); value = pileup phase only with the BAM already decoded in RAM" % (what, workers),
        "rows_per_s": rows / pileup_s, "positions": positions, "rows": rows,
        "end_to_end": {"positions_per_s": positions / total_s, "rows_per_s": rows / total_s, "total_s": total_s, "load_s": load_s, "threshold_s": thr_s, "pileup_s": pileup_s, "wall_s": wall},
    }


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; the JSON says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", choices=["full", "region"], default="full", help="CPU baseline + parity on the whole bench BAM (default) or on its first eighth")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)  # the short run the --pmc passes profile
    ap.add_argument("--tile", type=int, default=0, help="experiments: reference positions per accumulate tile (0 = the library's plan)")
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
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    contig, full_len, full_reads, gflags, needs_ref, desc = WORKLOADS[a.workload]
    contig_len, n_reads = int(full_len * a.scale), int(full_reads * a.scale)
    tmp = os.environ.get("MKP_BENCH_DIR", "/tmp")
    # the generator binary normally ships prebuilt (__graft_entry__.build()); if it has to be compiled, one rank does it
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "tools", "gen_modbam")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)
    if dist:
        dist.barrier()
    hemi = a.workload == "hemi"
    seed = {"c3": 20, "hemi": 30}.get(a.workload, 1) + rank
    t0 = time.time()
    bam, fa, meta = gen_bam(os.path.join(tmp, "mkp_%s_L%d_N%d_seed%d" % (a.workload, contig_len, n_reads, seed)), contig, contig_len, n_reads, seed, gflags,
                            max(1, (os.cpu_count() or 1) // world))
    gen_s = time.time() - t0
    # `-t 8`: the reference's --threads (8 here, its default is 4) sets how many intervals are in flight (chunk size floor(1.5 t)) and how
    # the sampling schedule batches its intervals; the device run and the CPU baseline get the same value so that they sample the same reads
    flags = (["--cpg", "--ref", fa] if needs_ref else []) + ["-t", "8"]

    def run_subcommand(out, extra):   # `modkit pileup` / `modkit pileup-hemi` on the bench context
        return ctx.pileup_hemi_run([bam, "-o", out] + flags + extra) if hemi else ctx.pileup_run([bam, out] + flags + extra)

    ctx = modkit_amd.Context(device=local_rank, tile_positions=a.tile)
    out_bed = bam + ".device.bed"
    rep = None
    if world == 1 and not (a.inner or a.skip_e2e):
        # end to end: the whole subcommand on this context (block reads + inflate, threshold sampling, focus, device pipeline,
        # bedMethyl text), with the driver's default sharding (the next shard's blocks inflate while this one is packed and run)
        rep = run_subcommand(out_bed, [])
        thr_h = [float(rep.threshold[i]) if rep.has_threshold[i] else 0.0 for i in range(4)]
    elif world == 1:
        thr = ctx.estimate_thresholds(bam, ["-t", "8"])
        thr_h = [float(thr.get(b, 0.0)) for b in "ACGT"]
    else:
        # per-base thresholds from ALL ranks' samples: per-rank histograms of the sampled probabilities, summed over RCCL
        from modkit_amd import distributed as mkd
        thr = mkd.estimate_thresholds_allreduce(ctx, bam, [])   # every rank samples its own BAM in full; histograms summed over RCCL
        thr_h = [float(thr.get(b, 0.0)) for b in "ACGT"]
    # kernels-only tier: the whole contig as ONE HBM-resident shard (same thresholds), re-launched K times
    targv = []
    for i in range(4):
        if thr_h[i] > 0:
            targv += ["--filter-threshold", "%s:%r" % ("ACGT"[i], thr_h[i])]
    rep1 = run_subcommand(out_bed + ".oneshard", targv + ["--shard-bytes", str(1 << 40)])
    if rep is None:
        rep = rep1
    elif sh256(out_bed) != sh256(out_bed + ".oneshard"):
        raise SystemExit("sharded and single-shard bedMethyl differ")
    os.remove(out_bed + ".oneshard")
    n_rows = int(rep.n_rows)
    ctx.rerun(a.warmup)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.rerun(a.steps)  # K passes; each launch sequence ends with a stream sync inside the library
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    if a.inner:
        ctx.close()
        return
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    totals = torch.tensor([float(contig_len), float(n_rows)], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    total_positions, total_rows = float(totals[0].item()), float(totals[1].item())

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        kernels = {"mkp_decode_*": (st.decode_kernel_ms, st.alg_bytes_decode), "mkp_pileup_tiles": (st.pileup_kernel_ms, st.alg_bytes_pileup)}
        if st.rows_kernel_ms > 0:
            kernels["mkp_emit_rows"] = (st.rows_kernel_ms, st.alg_bytes_rows)
        # the roofline is reported for the aggregation kernel (north_star's target), whichever kernel is slowest; its name in the
        # rocprof summaries: mkp_pileup_tiles_focus for runs with focus positions (--cpg), mkp_pileup_tiles otherwise
        dom = "mkp_pileup_tiles"
        dom_kernel = "mkp_pileup_tiles_hemi" if hemi else "mkp_pileup_tiles_focus" if needs_ref else "mkp_pileup_tiles"
        dom_ms, dom_bytes = kernels[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        slowest = max(kernels, key=lambda k: kernels[k][0])
        traffic, traffic_src, traffic_all = None, None, None
        if world == 1 and not a.no_pmc:
            tail = ["--workload", a.workload, "--scale", str(a.scale), "--no-cpu-baseline", "--no-pmc"]
            traffic_all, traffic_src = pmc_traffic(tail)
            if traffic_all:
                traffic = traffic_all.get(dom_kernel)
                traffic_all["mkp_pileup_tiles"] = traffic
                traffic_all["mkp_decode_*"] = sum(v for k, v in traffic_all.items() if k.startswith("mkp_decode_") and ":" not in k) or None
        dev_ms = rep1.pack_ms + rep1.h2d_ms + rep1.kernel_ms + rep1.d2h_ms
        result = {
            "metric": "genomic positions/sec pileup (bedMethyl rows/s); bit-exact vs ref", "value": total_positions * a.steps / elapsed, "unit": "positions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": (desc % (contig_len, n_reads, meta["aligned_bases"] / max(1, meta["reads"]), meta["aligned_bases"] / contig_len)) + ("; one such shard (own contig) per GPU, thresholds all-reduced over RCCL" if world > 1 else ""),
                       "scale": a.scale, "rows_per_s": total_rows * a.steps / elapsed, "rows_per_step": total_rows, "reads": int(st.n_reads), "call_events": int(st.n_events),
                       "tiles": int(st.n_tiles), "thresholds": {"ACGT"[i]: thr_h[i] for i in range(4) if thr_h[i] > 0},
                       "kernel_ms": {"decode": st.decode_kernel_ms, "pileup": st.pileup_kernel_ms, "rows": st.rows_kernel_ms, "gather": st.gather_kernel_ms},
                       "generator_s": gen_s, "arithmetic": "u32 tallies in LDS; f32 threshold caller (bit-exact vs the reference's f32)",
                       "parity": "bit-exact vs the restated CPU path (oracle/) on this BAM; the oracle is pinned on the reference's golden files; ties / >=3 codes / QC-fail / N ops are reference-unpinned (DESIGN.md §7)",
                       "roofline_all_kernels": {k: {"achieved_GBps": v[1] / (v[0] * 1e-3) / 1e9, "algorithmic_bytes": int(v[1]), "avg_launch_ms": v[0], "frac": v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                    "traffic": (traffic_all or {}).get(k)} for k, v in kernels.items() if v[0] > 0},
                       "slowest_kernel": slowest},
            "tiers": {
                "kernels_only": {"positions_per_s": total_positions * a.steps / elapsed, "rows_per_s": total_rows * a.steps / elapsed, "ms": ms_per_step, "what": "timed region: K re-launches on the HBM-resident shard"},
                "device_pipeline": {"positions_per_s": rep1.n_positions / (dev_ms * 1e-3), "rows_per_s": rep1.n_rows / (dev_ms * 1e-3), "ms": dev_ms,
                                    "stages_ms": {"pack": rep1.pack_ms, "h2d": rep1.h2d_ms, "kernels": rep1.kernel_ms, "d2h": rep1.d2h_ms}, "what": "rank 0, the contig as one shard, first (cold) pass: host pack + H2D + kernels + D2H of rows"},
                "end_to_end": {"positions_per_s": rep.n_positions / (rep.total_ms * 1e-3), "rows_per_s": rep.n_rows / (rep.total_ms * 1e-3), "ms": rep.total_ms,
                               "stages_ms": {"bam_load_inflate": rep.load_ms, "threshold": rep.threshold_ms, "focus": rep.focus_ms, "pack": rep.pack_ms, "h2d": rep.h2d_ms, "kernels": rep.kernel_ms,
                                             "d2h": rep.d2h_ms, "bedmethyl_text_write": rep.write_ms},
                               "shards": int(rep.n_shards), "what": "rank 0: mkp_pileup_run wall (`modkit pileup in.bam out.bed` with the workload's flags, default sharding), page cache warm; bam_load_inflate = what the shard loop waited for blocks (the rest overlaps with pack / run / write)"},
            },
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_read": (traffic_all or {}).get(dom_kernel + ":read"), "traffic_write": (traffic_all or {}).get(dom_kernel + ":write"), "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": dom_ms},
        }
        if world == 1 and not a.no_cpu_baseline:
            workers = min(os.cpu_count() or 1, 8)
            obed, region, base = cpu_baseline(bam, flags, contig, contig_len, workers, a.cpu_sample, hemi)
            if region:
                dbed = bam + ".device.region.bed"
                if hemi:
                    modkit_amd.pileup_hemi([bam, "-o", dbed, "--device", str(local_rank)] + flags + region)
                else:
                    modkit_amd.pileup([bam, dbed, "--device", str(local_rank)] + flags + region)
            else:
                dbed = out_bed
            base["bedmethyl_sha256_equal"] = sh256(dbed) == sh256(obed)
            base["bedmethyl_sha256"] = sh256(dbed)
            base["speedup_end_to_end"] = (rep.n_positions / (rep.total_ms * 1e-3)) / base["end_to_end"]["positions_per_s"] if not region else None
            if (os.cpu_count() or 1) > 8:   # the same run on more of this box's cores (the reference's --threads is the user's choice)
                w2 = min(os.cpu_count(), 32)
                _, _, b2 = cpu_baseline(bam, [f for f in flags if f not in ("-t", "8")] + ["-t", str(w2)], contig, contig_len, w2, a.cpu_sample, hemi)   # (-t steers its sampling schedule too: timing only, no sha comparison)
                base["more_cores"] = {"cores": w2, "positions_per_s": b2["value"], "end_to_end": b2["end_to_end"],
                                      "speedup_end_to_end": (rep.n_positions / (rep.total_ms * 1e-3)) / b2["end_to_end"]["positions_per_s"] if not region else None}
            result["cpu_baseline"] = base
        print(json.dumps(result))
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
