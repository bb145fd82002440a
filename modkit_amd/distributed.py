"""Multi-GPU glue for libmkpileup: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The pileup path itself shards by reference windows with no data-path collective
(mkp_pileup_main --gpus-rank R --gpus-world W); the only thing ranks ever share is the pass threshold:

* broadcast_thresholds  — default sampled mode: rank 0 estimates (mkp_estimate_thresholds), everyone receives it.
* estimate_thresholds_allreduce / percentile_from_histograms — full-data percentile (`-f 1.0`, thresholds.rs:121-159) when
  every rank has decoded only its own windows: the sample stays in each GPU's HBM, two-level histograms of the f32 *bit
  patterns* (top 16 bits, then the low 16 bits inside the bins that hold the wanted order statistics) are summed across ranks
  (int64 all-reduce, 512 KiB each) and each rank evaluates percentile_linear_interp (thresholds.rs:17-38) — bit-identical to
  sorting the union of all values.
* pileup_sharded — the launcher: thresholds as above, then `mkp_pileup_main --gpus-rank R --gpus-world W` per rank and an
  ordered concatenation of the rank outputs.
"""
import ctypes

import numpy as np

from . import lib

BASES = "ACGT"


def _dist():
    import torch.distributed as dist
    return dist


def _device():
    import torch
    dist = _dist()
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_thresholds(thresholds, src=0):
    """thresholds: {base letter: f32} on `src` (ignored elsewhere) -> the same dict on every rank."""
    import torch
    dist = _dist()
    t = torch.zeros(8, dtype=torch.float32, device=_device())
    if not dist.is_initialized() or dist.get_rank() == src:
        for b, v in (thresholds or {}).items():
            i = BASES.index(b)
            t[i] = float(np.float32(v))
            t[4 + i] = 1.0
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    h = t.cpu().numpy()
    return {BASES[i]: float(h[i]) for i in range(4) if h[4 + i] > 0}


def _u64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def _allreduce_u64(h):
    """SUM all-reduce of a uint64 numpy array over the ranks (int64 on the wire: counts stay far below 2^63)."""
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return h
    t = torch.from_numpy(h.astype(np.int64)).to(_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def percentile_from_histograms(get, q):
    """percentile_linear_interp (thresholds.rs:17-38) of the union of all ranks' samples of one base.
    get(level, prefix) -> this rank's uint64[65536] histogram (Context.histogram_get or mkp_histogram_from_values);
    every call is followed by a SUM all-reduce, so all ranks must call this function in the same order.
    Returns (value or None when the union is empty, n)."""
    L = lib()
    h0 = _allreduce_u64(np.ascontiguousarray(get(0, 0), dtype=np.uint64))
    n = int(h0.sum())
    if n == 0:
        return None, 0
    bins = (ctypes.c_uint32 * 2)()
    rk = (ctypes.c_uint64 * 2)()
    nn = ctypes.c_uint64()
    if L.mkp_histogram_locate(_u64p(h0), ctypes.c_float(q), bins, rk, ctypes.byref(nn)) != 0:
        raise ValueError("not enough datapoints, got %d" % n)
    ys = []
    h1 = None
    for k in range(2):
        if k == 0 or bins[1] != bins[0]:
            h1 = _allreduce_u64(np.ascontiguousarray(get(1, int(bins[k])), dtype=np.uint64))
        y = ctypes.c_float()
        if L.mkp_histogram_resolve(int(bins[k]), _u64p(h1), int(rk[k]), ctypes.byref(y)) != 0:
            raise ValueError("histogram levels disagree")
        ys.append(y.value)
    out = ctypes.c_float()
    if L.mkp_percentile_from_histogram(n, ctypes.c_float(q), ctypes.c_float(ys[0]), ctypes.c_float(ys[1]), ctypes.byref(out)) != 0:
        raise ValueError("not enough datapoints, got %d" % n)
    return float(out.value), n


def host_histogram(values):
    """get(level, prefix) over f32 values held on the host (mkp_histogram_from_values): for callers that sampled elsewhere, and tests."""
    v = np.ascontiguousarray(values, dtype=np.float32)

    def get(level, prefix):
        out = np.zeros(65536, dtype=np.uint64)
        rc = lib().mkp_histogram_from_values(v.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(v), level, prefix, _u64p(out))
        if rc != 0:
            raise ValueError("mkp_histogram_from_values failed (%d)" % rc)
        return out
    return get


def estimate_thresholds_allreduce(ctx, bam, flags=(), q=0.1, rank=None, world=None):
    """Per-base pass thresholds over the samples of ALL ranks (full-data mode, `-f 1.0`): each rank decodes only the reads of its
    own sampling intervals on its GPU (mkp_histogram_add_bam --gpus-rank R --gpus-world W), the two-level histograms are summed
    over the ranks (RCCL all-reduce on GPUs, gloo in the CPU tests) and every rank evaluates the exact percentile of the union.
    With rank/world given explicitly the ranks shard ONE BAM; by default (torch.distributed rank/world when initialised and each
    rank holding its own BAM, as in bench.py) every rank samples its whole file."""
    dist = _dist()
    shard = rank is not None and world is not None and world > 1
    ctx.histogram_begin()
    argv = list(flags) + ["-f", "1.0", "-p", repr(float(q))]
    if shard:
        argv += ["--gpus-rank", str(rank), "--gpus-world", str(world)]
    ctx.histogram_add_bam(bam, argv)
    out = {}
    for b in BASES:
        t, n = percentile_from_histograms(lambda level, prefix, b=b: ctx.histogram_get(b, level, prefix), q)
        if t is not None:
            out[b] = t
    return out


SAMPLING_FLAGS_WITH_VALUE = ("--region", "--sample-region", "--include-bed", "--include-positions", "--edge-filter", "--ignore", "--preset",
                             "--sampling-interval-size", "-t", "--threads")
SAMPLING_FLAGS_BARE = ("--include-unmapped", "--invert-edge-filter")


def pileup_sharded(argv, rank=None, world=None, device=None, q=0.1, stats=None):
    """`modkit pileup` with ONE BAM sharded over the ranks of the current torch.distributed job (one process per GPU):
    thresholds from the all-reduced histograms of rank-sharded sampling, then every rank runs its contiguous run of the
    reference's interval grid (mkp_pileup_main --gpus-rank/--gpus-world) into `<out>.rank<R>`, and rank 0 concatenates the parts
    in rank order into `<out>`.  argv = [in.bam, out.bed, flags...] (no threshold flags).
    Parity: the thresholds are the FULL-DATA percentile (`-f 1.0`; the count-based default sampler carries quotas from interval
    to interval and does not shard), so the output is byte-identical to a single-GPU run with `-f 1.0 -p q` — or with the explicit
    `--filter-threshold` values this function returns — not to a default (`-n 10042`) run.
    `--with-header` is written by rank 0 only; `--bgzf` (one BGZF stream + one index) and `--partition-tag` (one file per key)
    do not concatenate and are refused.  `stats` (a dict) receives this rank's wall times."""
    import os
    import shutil
    from . import Context, pileup
    dist = _dist()
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    bam, out, flags = argv[0], argv[1], list(argv[2:])
    for bad in ("--bgzf", "--partition-tag", "--bedgraph"):
        if bad in flags:
            raise ValueError("pileup_sharded: %s output does not concatenate over ranks; run it on one GPU" % bad)
    if rank > 0:   # one header, from rank 0
        flags = [f for f in flags if f not in ("--with-header", "--header")]
    sflags, i = [], 0
    while i < len(flags):
        if flags[i] in SAMPLING_FLAGS_WITH_VALUE:
            sflags += flags[i:i + 2]
            i += 2
        else:
            if flags[i] in SAMPLING_FLAGS_BARE:
                sflags.append(flags[i])
            i += 1
    import time
    t0 = time.time()
    ctx = Context(device=0 if device is None else device)
    try:
        thr = estimate_thresholds_allreduce(ctx, bam, sflags, q=q, rank=rank, world=world)
    finally:
        ctx.close()
    t1 = time.time()
    targv = []
    for b, v in sorted(thr.items()):
        targv += ["--filter-threshold", "%s:%r" % (b, v)]
    if not targv:
        raise ValueError("no mod calls sampled on any rank")
    part = "%s.rank%d" % (out, rank)
    pileup([bam, part] + flags + targv + ["--gpus-rank", str(rank), "--gpus-world", str(world)] + (["--device", str(device)] if device is not None else []))
    if stats is not None:
        stats.update({"threshold_s": t1 - t0, "pileup_s": time.time() - t1, "part_bytes": os.path.getsize(part)})
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        with open(out, "wb") as f:
            for r in range(world):
                with open("%s.rank%d" % (out, r), "rb") as g:
                    shutil.copyfileobj(g, f)
                os.remove("%s.rank%d" % (out, r))
    if dist.is_initialized():
        dist.barrier()
    return thr


def shard_plan(argv, rank, world):
    """The reference windows `mkp_pileup_main argv --gpus-rank rank --gpus-world world` would process, without touching
    a device (host-side scheduling only): list of (contig, start, end)."""
    import os
    import tempfile
    from . import pileup
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "plan.tsv")
        pileup([argv[0], out] + list(argv[2:]) + ["--plan-only", "--gpus-rank", str(rank), "--gpus-world", str(world)])
        return [(l[0], int(l[1]), int(l[2])) for l in (x.split("\t") for x in open(out).read().splitlines())]
