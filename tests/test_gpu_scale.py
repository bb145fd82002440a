"""GPU parity, part 3: the BASELINE.json configs.  Scaled-down C2..C5 against the oracle (whole bedMethyl text,
bit-exact); the full-size C2 and C3 workloads against the oracle (sha256 of the whole bedMethyl); and the full-size C2
workload through size-independent properties: row invariants, idempotence of a re-run on the resident shard, and
shard-split invariance (two half-contig shards == one whole-contig shard)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import modkit_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gen(tmp, name, contigs, reads, style, seed, extra=()):
    tool = os.path.join(ROOT, "tools", "gen_modbam")
    if not os.path.exists(tool):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)
    prefix = os.path.join(str(tmp), name)
    args = [tool, "--out", prefix, "--reads", str(reads), "--seed", str(seed), "--style", style] + list(extra)
    for c in contigs:
        args += ["--contig", "%s:%d" % c]
    meta = json.loads(subprocess.check_output(args).decode())
    return prefix + ".bam", prefix + ".fa", meta


def both(oracle_bin, tmp, bam, flags):
    dev, ora = os.path.join(str(tmp), "dev.bed"), os.path.join(str(tmp), "ora.bed")
    modkit_amd.pileup([bam, dev] + flags)
    p = subprocess.run([oracle_bin, "pileup", bam, ora, "--oracle-workers", "8"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-400:]
    a, b = open(dev).read(), open(ora).read()
    if a != b:
        al, bl = a.splitlines(), b.splitlines()
        for i in range(max(len(al), len(bl))):
            x, y = (al[i] if i < len(al) else "<none>"), (bl[i] if i < len(bl) else "<none>")
            assert x == y, "row %d differs\n device: %s\n oracle: %s (%d vs %d rows)" % (i, x, y, len(al), len(bl))
    return a


def test_c2_scaled_defaults(oracle_bin, tmp_path):
    # C2 at 1/10: one contig, `C+m?` on every read CpG, default sampled 10th-percentile threshold
    bam, fa, meta = gen(tmp_path, "c2", [("synth5m", 500_000)], 10_000, "m", 11)
    out = both(oracle_bin, tmp_path, bam, [])
    assert len(out.splitlines()) > 100_000


def test_c3_scaled_cpg_hm_ties(oracle_bin, tmp_path):
    # C3 scaled: CpG-depleted contig, mean-10 kb reads, `C+hm?` / `C+h?;C+m?` alternating with forced h==m ties, --cpg --ref,
    # default -i 100000 (boundary CpGs are lost exactly as the reference loses them)
    bam, fa, meta = gen(tmp_path, "c3", [("chr20", 1_000_000)], 3_000, "hm", 20, ["--cpg-depleted", "--mean-len", "10000"])
    out = both(oracle_bin, tmp_path, bam, ["--cpg", "--ref", fa])
    assert len(out.splitlines()) > 10_000


def test_c4_scaled_traditional_two_ranks(oracle_bin, tmp_path):
    # C4 scaled: several contigs, --preset traditional; interval-sharded over 2 ranks (run back to back on this GPU):
    # concatenating the rank outputs in rank order reproduces the single-GPU bedMethyl and the oracle's
    bam, fa, meta = gen(tmp_path, "c4", [("chr1", 400_000), ("chr2", 300_000), ("chrX", 150_000)], 4_000, "hm", 4, ["--cpg-depleted", "--mean-len", "8000"])
    flags = ["--preset", "traditional", "--ref", fa, "-f", "1.0", "-p", "0.1"]
    whole = both(oracle_bin, tmp_path, bam, flags)
    parts = []
    for r in range(2):
        o = os.path.join(str(tmp_path), "rank%d.bed" % r)
        modkit_amd.pileup([bam, o] + flags + ["--gpus-rank", str(r), "--gpus-world", "2"])
        parts.append(open(o).read())
    assert all(parts) and "".join(parts) == whole


def test_c5_scaled_multimod_bed(oracle_bin, tmp_path):
    # C5 scaled: `C+h?;C+m?;A+a?` (6mA on every A), per-mod thresholds on top of estimated per-base ones, --include-bed
    bam, fa, meta = gen(tmp_path, "c5", [("chr1", 200_000), ("chr2", 100_000)], 1_500, "hma", 5, ["--mean-len", "6000"])
    rng = np.random.default_rng(5)
    bed = os.path.join(str(tmp_path), "inc.bed")
    with open(bed, "w") as f:
        for i in range(60):
            c, ln = ("chr1", 200_000) if i % 3 else ("chr2", 100_000)
            s = int(rng.integers(0, ln - 2000))
            kind = i % 4
            f.write("%s\t%d\t%d\n" % (c, s, s + 2000) if kind == 0 else "%s\t%d\t%d\tx\t0\t%s\n" % (c, s, s + 2000, "+-."[kind - 1]))
    out = both(oracle_bin, tmp_path, bam, ["--mod-thresholds", "m:0.8", "--mod-thresholds", "h:0.9", "--mod-thresholds", "a:0.7", "--include-bed", bed])
    assert len(out.splitlines()) > 10_000


HG38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973),
        ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718),
        ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983),
        ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]


def test_c4_genome_scale_model_vs_oracle(oracle_bin, tmp_path):
    # BASELINE configs[3] as a declared scale model: all 24 hg38 contigs at 1/100 of their lengths (31 Mb), 30x, --preset traditional
    # --ref; the default sampled threshold and the full-data percentile (-f 1.0), whole output sha256 against the oracle
    contigs = [(n, l // 100) for n, l in HG38]
    total = sum(l for _, l in contigs)
    bam, fa, meta = gen(tmp_path, "c4g", contigs, int(30 * total / 9994), "hm", 40, ["--cpg-depleted", "--mean-len", "8353"])
    assert meta["aligned_bases"] > 25 * total
    n = _full_vs_oracle(oracle_bin, tmp_path, bam, ["--preset", "traditional", "--ref", fa], 250_000)
    assert n == _full_vs_oracle(oracle_bin, tmp_path, bam, ["--preset", "traditional", "--ref", fa, "-f", "1.0"], 250_000)


def test_c5_genome_scale_model_vs_oracle(oracle_bin, tmp_path):
    # BASELINE configs[4] as a declared scale model: 24 contigs at 1/400 (7.7 Mb), 60x, C+h?;C+m?;A+a? (6mA at every A), per-mod thresholds over
    # estimated per-base ones, --include-bed = seeded random 2 kb intervals at the full genome's density (seed 5, mixed BED3 / BED6)
    contigs = [(n, max(100_000, l // 400)) for n, l in HG38]
    total = sum(l for _, l in contigs)
    bam, fa, meta = gen(tmp_path, "c5g", contigs, int(60 * total / 9994), "hma", 50, ["--cpg-depleted", "--mean-len", "8353"])
    import random
    rng = random.Random(5)
    bed = os.path.join(str(tmp_path), "inc.bed")
    with open(bed, "w") as f:
        for i in range(50):
            x = rng.randrange(total)
            for name, ln in contigs:
                if x < ln:
                    break
                x -= ln
            s = max(0, min(x, ln - 2000)); kind = rng.randrange(4)
            f.write("%s\t%d\t%d\n" % (name, s, s + 2000) if kind == 0 else "%s\t%d\t%d\tiv%d\t0\t%s\n" % (name, s, s + 2000, i, "+-."[kind - 1]))
    _full_vs_oracle(oracle_bin, tmp_path, bam, ["--mod-thresholds", "m:0.8", "--mod-thresholds", "h:0.9", "--mod-thresholds", "a:0.7", "--include-bed", bed], 50_000)


def _digest(r):
    h = hashlib.sha256()
    for f in modkit_amd.ROW_FIELDS:
        h.update(np.ascontiguousarray(r[f]).tobytes())
    return h.hexdigest()


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def _full_vs_oracle(oracle_bin, tmp_path, bam, flags, min_rows):
    """whole bedMethyl of the full-size workload, device vs oracle: sha256 (and the first differing row if they differ)"""
    dev, ora = os.path.join(str(tmp_path), "dev.bed"), os.path.join(str(tmp_path), "ora.bed")
    modkit_amd.pileup([bam, dev] + flags)
    p = subprocess.run([oracle_bin, "pileup", bam, ora, "--oracle-workers", str(min(os.cpu_count() or 1, 16))] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-400:]
    if _sha(dev) != _sha(ora):
        with open(dev) as fa, open(ora) as fb:
            for i, (x, y) in enumerate(zip(fa, fb)):
                assert x == y, "row %d differs\n device: %s oracle: %s" % (i, x, y)
        assert os.path.getsize(dev) == os.path.getsize(ora), "one output is a prefix of the other"
    n = sum(1 for _ in open(dev))
    assert n > min_rows
    return n


def _relaunch_vs_oracle(tmp_path, bam, flags):
    """The kernels the bench times are the kernels whose rows are checked: `modkit pileup` on a context with the whole contig as ONE resident
    shard (file == the oracle's, sha256), then the row columns left in HBM by that one-shot launch and by two RE-LAUNCHES on the resident
    shard (mkp_shard_rerun, what bench.py's timed region calls) == the columns of the ORACLE's bedMethyl.  (_full_vs_oracle left ora.bed.)"""
    ora, dev = os.path.join(str(tmp_path), "ora.bed"), os.path.join(str(tmp_path), "dev_resident.bed")
    want = modkit_amd.rows_digest(modkit_amd.read_bedmethyl(ora))
    ctx = modkit_amd.Context(device=0)
    try:
        rep = ctx.pileup_run([bam, dev] + flags + ["--shard-bytes", str(1 << 40)])
        assert rep.n_shards == 1 and _sha(dev) == _sha(ora)
        one_shot = modkit_amd.rows_to_numpy(ctx.rerun(0, fetch=True))     # no launch: the rows of the shard pass itself
        assert modkit_amd.rows_digest(one_shot) == want
        again = modkit_amd.rows_to_numpy(ctx.rerun(2, fetch=True))        # two re-launches, rows of the second
        assert modkit_amd.rows_digest(again) == want
        assert np.array_equal(again["motif_idx"], one_shot["motif_idx"])
    finally:
        ctx.close()


def test_c2_full_size_vs_oracle(oracle_bin, tmp_path):
    # BASELINE configs[1] at full size: 5 Mb contig, 100 000 reads (~96x), C+m?, defaults (sampled 10th-percentile threshold)
    bam, fa, meta = gen(tmp_path, "c2full", [("synth5m", 5_000_000)], 100_000, "m", 1)
    _full_vs_oracle(oracle_bin, tmp_path, bam, [], 1_500_000)
    _relaunch_vs_oracle(tmp_path, bam, [])


def test_c3_full_size_vs_oracle(oracle_bin, tmp_path):
    # BASELINE configs[2] at full size (the bench workload): chr20-sized contig, 193 000 reads (~30x), C+hm? / C+h?;C+m?, --cpg --ref
    bam, fa, meta = gen(tmp_path, "c3full", [("chr20", 64_444_167)], 193_000, "hm", 20, ["--cpg-depleted", "--mean-len", "8353"])
    assert meta["aligned_bases"] > 1_800_000_000
    _full_vs_oracle(oracle_bin, tmp_path, bam, ["--cpg", "--ref", fa], 2_000_000)
    _relaunch_vs_oracle(tmp_path, bam, ["--cpg", "--ref", fa])


def test_c2_full_size_properties(tmp_path):
    # the bench workload itself: 5 Mb contig, 100 000 reads (~96x)
    bam, fa, meta = gen(tmp_path, "c2full", [("synth5m", 5_000_000)], 100_000, "m", 1)
    ctx = modkit_amd.Context(device=0)
    try:
        ctx.set_caller(per_base={"C": 0.802734375})
        whole = modkit_amd.rows_to_numpy(ctx.process_region(bam, 0, 0, 5_000_000))
        n = len(whole["pos"])
        assert n > 1_500_000
        # rows ordered by (position, strand '+' < '-'); one mod code here
        key = whole["pos"].astype(np.int64) * 2 + (whole["strand"] == ord("-"))
        assert np.all(np.diff(key) > 0)
        # PileupFeatureCounts invariants (pileup/mod.rs:283-410)
        assert np.array_equal(whole["n_valid"], whole["n_mod"] + whole["n_canonical"] + whole["n_other"])
        assert np.all(whole["n_valid"] > 0) and np.all(whole["n_other"] == 0) and np.all(whole["code_repr"] == ord("m"))
        depth = whole["n_valid"].astype(np.int64) + whole["n_fail"] + whole["n_diff"] + whole["n_nocall"] + whole["n_delete"]
        assert depth.max() <= 8000 and 20 < depth.mean() < 80  # per-strand column depth of a ~96x pileup
        # idempotence: re-running the resident shard reproduces the rows bit for bit
        again = modkit_amd.rows_to_numpy(ctx.rerun(2, fetch=True))
        assert _digest(again) == _digest(whole)
        # shard-split invariance: rows are independent of where the shard boundary falls
        left = modkit_amd.rows_to_numpy(ctx.process_region(bam, 0, 0, 2_345_678))
        right = modkit_amd.rows_to_numpy(ctx.process_region(bam, 0, 2_345_678, 5_000_000))
        joined = {f: np.concatenate([left[f], right[f]]) for f in modkit_amd.ROW_FIELDS}
        assert _digest(joined) == _digest(whole)
    finally:
        ctx.close()


def _sharded_worker(rank, world, port, argv, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both processes share this box's one GPU: RCCL needs one device per rank
    try:
        from modkit_amd import distributed as mkd
        thr = mkd.pileup_sharded(argv, device=0)
        q.put((rank, thr))
    finally:
        dist.destroy_process_group()


def test_c4_scaled_two_processes_sharded_sampling_allreduce(oracle_bin, tmp_path):
    # C4 scaled, two processes: each rank samples ONLY its own sampling intervals on the GPU (-f 1.0), the two-level histograms
    # are summed over the ranks (torch.distributed all-reduce; gloo here because both ranks sit on one GPU, RCCL on a node),
    # each rank runs its run of the interval grid, rank 0 concatenates: byte-identical to the single-rank run and to the oracle
    import socket
    import torch.multiprocessing as mp
    bam, fa, meta = gen(tmp_path, "c4s", [("chr1", 2_600_000), ("chr2", 1_700_000), ("chrX", 900_000)], 9_000, "hm", 44, ["--cpg-depleted", "--mean-len", "8000"])
    flags = ["--preset", "traditional", "--ref", fa, "--sampling-interval-size", "400000"]
    whole = both(oracle_bin, tmp_path, bam, flags + ["-f", "1.0", "-p", "0.1"])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.path.join(str(tmp_path), "sharded.bed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, [bam, out] + flags + ["-f", "1.0", "-p", "0.1"], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == res[1] and "C" in res[0]
    assert open(out).read() == whole


@pytest.mark.parametrize("extra", [[], ["-n", "800", "-p", "0.2"], ["-f", "0.4"]])
def test_two_processes_default_sampled_thresholds_equal_single_gpu(oracle_bin, tmp_path, extra):
    # `modkit pileup` defaults on two ranks: the count-based estimate does not shard — rank 0 walks the schedule (interval heads), the
    # thresholds are broadcast, every rank runs its windows: byte-identical to the single-GPU run with the same flags (and to the oracle)
    import socket
    import torch.multiprocessing as mp
    bam, fa, meta = gen(tmp_path, "c4d", [("chr1", 1_900_000), ("chr2", 1_200_000), ("chrX", 600_000)], 7_000, "hm", 45, ["--cpg-depleted", "--mean-len", "6000"])
    flags = ["--cpg", "--ref", fa, "--sampling-interval-size", "300000"] + extra
    whole = both(oracle_bin, tmp_path, bam, flags)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.path.join(str(tmp_path), "sharded.bed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, [bam, out] + flags, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == res[1] and "C" in res[0]
    assert open(out).read() == whole


def _dev_vs(text, tmp, bam, flags, name):
    out = os.path.join(str(tmp), name)
    p = subprocess.run([os.path.join(ROOT, "modkit_amd", "csrc", "mkpileup"), "pileup", bam, out] + flags + ["--stats"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-600:]
    assert open(out).read() == text, name
    return p.stderr


def test_shards_ahead_as_pieces_of_contigs_and_under_a_budget(oracle_bin, tmp_path):
    """Round 5: the shards of a run are ingested ahead of the loop whatever their cut — whole contigs, or pieces of contigs (a contig larger
    than a shard; here forced with --shard-bp) — several in flight under an HBM budget, and the full-data estimate (-f 1.0) samples from them
    in HBM: a read that lies in two pieces belongs to the first.  Every variant must write what the default run writes (== the oracle)."""
    bam, fa, meta = gen(tmp_path, "pc", [("chr1", 2_300_000), ("chr2", 1_500_000), ("chrX", 700_000)], 8_000, "hm", 47, ["--cpg-depleted", "--mean-len", "7000"])
    flags = ["--cpg", "--ref", fa, "--sampling-interval-size", "350000"]
    full = both(oracle_bin, tmp_path, bam, flags + ["-f", "1.0"])                          # whole contigs ahead, sampled from HBM
    err = _dev_vs(full, tmp_path, bam, flags + ["-f", "1.0", "--shard-bp", "450000"], "pieces_full.bed")   # pieces of contigs, sampled from HBM
    assert "resident_sampling=1" in err and "full-data threshold sample taken from" in err and "ahead=" in err and "ahead=0 " not in err and "ahead=3 " not in err, err[-400:]
    err = _dev_vs(full, tmp_path, bam, flags + ["-f", "1.0", "--shard-bp", "450000", "--hbm-budget-mb", "1"], "pieces_full_budget.bed")   # too small to hold them: host sampler, streamed shards
    assert "resident_sampling=0" in err, err[-400:]
    dflt = both(oracle_bin, tmp_path, bam, flags)                                          # the count-based default
    err = _dev_vs(dflt, tmp_path, bam, flags + ["--hbm-budget-mb", "1"], "default_budget.bed")           # whole contigs, one admitted at a time
    assert "resident_sampling=0" in err, err[-400:]
    err = _dev_vs(dflt, tmp_path, bam, flags + ["--shard-bp", "450000"], "pieces_default.bed")           # pieces: estimate from the host reader, then the pieces ahead of the loop
    assert "ahead=" in err and "ahead=0 " not in err, err[-400:]


@pytest.mark.parametrize("extra,run_extra", [(["-f", "1.0", "-p", "0.1"], ["--shard-bp", "400000"]), ([], ["--shard-bp", "400000", "--hbm-budget-mb", "1"])],
                         ids=["full_resident_pieces", "sampled_streamed"])
def test_two_processes_several_shards_per_rank(oracle_bin, tmp_path, extra, run_extra):
    """pileup_sharded = one mkp_pileup_run_cb per rank: each rank's shards (here several pieces per rank) are ingested ahead; in the full-data
    mode the rank samples them from HBM and the callback all-reduces the histograms, in the count-based mode the callback waits for rank 0's
    broadcast while the shards come in.  Byte-identical to the single-GPU run and the oracle."""
    import socket
    import torch.multiprocessing as mp
    bam, fa, meta = gen(tmp_path, "c4m", [("chr1", 2_100_000), ("chr2", 1_300_000), ("chrX", 800_000)], 7_500, "hm", 48, ["--cpg-depleted", "--mean-len", "6500"])
    flags = ["--preset", "traditional", "--ref", fa, "--sampling-interval-size", "300000"] + extra
    whole = both(oracle_bin, tmp_path, bam, flags)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.path.join(str(tmp_path), "sharded.bed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, [bam, out] + flags + run_extra, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == res[1] and "C" in res[0]
    assert open(out).read() == whole


def test_histogram_allreduce_through_rccl_on_one_rank(tmp_path):
    """mkp_histogram_allreduce: the device histograms widened to u64 and summed with ncclAllReduce on a communicator made through
    librccl's C API (modkit_amd.distributed.RcclComm).  One rank here (the box has one GPU): the sum is the histogram itself; what is
    checked is the call path — communicator, u64 device buffers, stream order, D2H — for both levels."""
    import numpy as np
    from modkit_amd import distributed as mkd
    bam, fa, meta = gen(tmp_path, "rc", [("chr1", 900_000)], 3_000, "hm", 46, ["--mean-len", "5000"])
    ctx = modkit_amd.Context()
    try:
        ctx.histogram_begin()
        ctx.histogram_add_bam(bam, ["-f", "1.0"])
        comm = mkd.RcclComm(rank=0, world=1)
        try:
            for b in "AC":
                h0 = ctx.histogram_get(b, 0, 0)
                r0 = ctx.histogram_allreduce(comm, b, 0, 0)
                assert np.array_equal(h0, r0)
                if h0.sum():
                    top = int(np.argmax(h0))
                    assert np.array_equal(ctx.histogram_get(b, 1, top), ctx.histogram_allreduce(comm, b, 1, top))
            assert ctx.histogram_get("C", 0, 0).sum() > 1000
            t_direct, n1 = mkd.percentile_from_histograms(lambda level, prefix: ctx.histogram_allreduce(comm, "C", level, prefix), 0.1, reduced=True)
            t_host, n2 = mkd.percentile_from_histograms(lambda level, prefix: ctx.histogram_get("C", level, prefix), 0.1)
            assert n1 == n2 and t_direct == t_host
        finally:
            comm.close()
    finally:
        ctx.close()
