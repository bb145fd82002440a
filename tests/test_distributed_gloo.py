"""N>1 path on CPU: world_size-2 gloo processes exercise what ranks share in a multi-GPU run — the shard plan
(disjoint, complete, ordered), the threshold broadcast and the histogram all-reduce for the full-data percentile
(thresholds.rs:17-38) — none of which needs a device."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "modkit_fixtures")


def percentile_linear_interp_f32(xs, q):
    """test-side restatement of percentile_linear_interp (src/thresholds.rs:17-38) in numpy float32"""
    xs = np.sort(np.asarray(xs, dtype=np.float32))
    q = np.float32(q)
    if q == np.float32(1.0):
        return np.float32(xs[-1])
    lq = np.float32(len(xs) - 1) * q
    left, right = xs[int(np.floor(lq))] , xs[int(np.ceil(lq))]
    g = np.float32(lq - np.float32(np.trunc(lq)))
    return np.float32(np.float32(left * np.float32(np.float32(1.0) - g)) + np.float32(right * g))


def _values(rank):
    rng = np.random.default_rng(100 + rank)
    q = rng.integers(0, 256, size=5000 + 777 * rank)
    v = ((q.astype(np.float32) + np.float32(0.5)) / np.float32(256.0)).astype(np.float32)  # quals_to_probs (mod_bam.rs:808-816)
    # a third of them as ReDistribute would leave them (p + m/2, m/3: not multiples of 1/512), so the low 16 pattern bits matter
    k = len(v) // 3
    v[:k] = (v[:k] / np.float32(3.0)).astype(np.float32)
    v[k:2 * k] = (v[k:2 * k] + v[:k] / np.float32(2.0)).astype(np.float32)
    return np.minimum(v, np.float32(1.0))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import modkit_amd
        from modkit_amd import distributed as mkd
        res = {}
        res["thr"] = mkd.broadcast_thresholds({"C": 0.802734375, "A": 0.5} if rank == 0 else None)
        # the path's one collective: two-level histograms of the f32 bit patterns, summed over the ranks (mkp_histogram_* C ABI)
        get = mkd.host_histogram(_values(rank))
        pn = [mkd.percentile_from_histograms(get, p) for p in (0.1, 0.25, 0.5, 0.999, 1.0, 0.0)]
        res["n"] = pn[0][1]
        res["pct"] = [t for t, _ in pn]
        bam = os.path.join(FIX, "bc_anchored_10_reads.sorted.bam")
        res["plan"] = mkd.shard_plan([bam, "-", "-i", "25", "--no-filtering"], rank, world)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope="module")
def two_ranks():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import modkit_amd
    modkit_amd.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_threshold_broadcast(two_ranks):
    assert two_ranks[0]["thr"] == two_ranks[1]["thr"] == {"A": 0.5, "C": 0.802734375}


def test_histogram_allreduce_percentile_equals_sorting_the_union(two_ranks):
    union = np.concatenate([_values(0), _values(1)])
    assert two_ranks[0]["n"] == two_ranks[1]["n"] == len(union)
    want = [float(percentile_linear_interp_f32(union, p)) for p in (0.1, 0.25, 0.5, 0.999, 1.0, 0.0)]
    assert two_ranks[0]["pct"] == want and two_ranks[1]["pct"] == want  # bit-exact f32


def test_shard_plans_partition_the_genome(two_ranks):
    sys.path.insert(0, ROOT)
    from modkit_amd import distributed as mkd
    bam = os.path.join(FIX, "bc_anchored_10_reads.sorted.bam")
    whole = mkd.shard_plan([bam, "-", "-i", "25", "--no-filtering"], 0, 1)
    a, b = two_ranks[0]["plan"], two_ranks[1]["plan"]
    assert a and b and not (set(a) & set(b))
    # ranks take contiguous runs in genome order: concatenating the per-rank plans reproduces the single-GPU order
    cover = lambda plan: sorted((c, p) for c, s, e in plan for p in range(s, e))
    assert cover(a + b) == cover(whole)
    order = {c: i for i, c in enumerate(dict.fromkeys(w[0] for w in whole))}
    assert [order[w[0]] for w in a + b] == sorted(order[w[0]] for w in a + b)
    assert max(order[w[0]] for w in a) <= min(order[w[0]] for w in b)


def test_threshold_mode_follows_the_flags():
    """pileup_sharded decides like the subcommand: thresholds given, the full-data percentile (`-f 1.0`: sharded sampling + all-reduce), or
    the default count-based estimate (rank 0 walks the schedule, four floats are broadcast); the estimate flags leave the sharded runs'
    argv, the sampling flags stay in both."""
    from modkit_amd.distributed import split_threshold_flags as sp
    s, e, rest, mode, q = sp(["--cpg", "--ref", "x.fa", "-t", "8"])
    assert mode == "sampled" and q == 0.1 and e == [] and s == ["-t", "8"] and rest == ["--cpg", "--ref", "x.fa", "-t", "8"]
    s, e, rest, mode, q = sp(["-n", "500", "-p", "0.25", "--ignore", "h", "--region", "chr1:1-20", "--include-unmapped"])
    assert mode == "sampled" and q == 0.25 and e == ["-n", "500", "-p", "0.25"] and s == ["--ignore", "h", "--region", "chr1:1-20", "--include-unmapped"]
    assert rest == ["--ignore", "h", "--region", "chr1:1-20", "--include-unmapped"]
    assert sp(["-f", "1.0"])[3] == "full" and sp(["--sampling-frac", "0.5"])[3] == "sampled"
    assert sp(["--filter-threshold", "C:0.8", "-f", "1.0"])[3] == "given" and sp(["--no-filtering"])[3] == "given"
    assert sp(["-f", "0.3"], mode="full")[3] == "full"
    with pytest.raises(ValueError):
        sp([], mode="median")
