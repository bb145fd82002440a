"""Multi-GPU glue for libmkpileup: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The pileup path itself shards by reference windows with no data-path collective
(mkp_pileup_main --gpus-rank R --gpus-world W); the only thing ranks ever share is the pass threshold:

* broadcast_thresholds  — default sampled mode: rank 0 estimates (mkp_estimate_thresholds), everyone receives it.
* allreduce_histograms + percentile_from_histogram — full-data percentile (`-f 1.0`, thresholds.rs:121-159) when every
  rank has decoded only its own windows: histograms keyed by the f32 *bit pattern* of the probabilities are summed
  across ranks (int64 all-reduce) and each rank evaluates percentile_linear_interp (thresholds.rs:17-38) on the merged
  histogram — bit-identical to sorting the union of all values.
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


def local_histogram(values):
    """f32 values -> (sorted distinct bit patterns as uint32, counts as int64).  Probabilities are positive, so the
    unsigned order of the bit patterns is the numeric order."""
    v = np.ascontiguousarray(values, dtype=np.float32).view(np.uint32)
    keys, counts = np.unique(v, return_counts=True)
    return keys.astype(np.uint32), counts.astype(np.int64)


def allreduce_histograms(values):
    """values: this rank's f32 probabilities for one canonical base -> (keys f32 ascending, counts int64) of the union
    over all ranks.  Two collectives: all-gather of the (few, <= 2^16) distinct keys, all-reduce(SUM) of the counts."""
    import torch
    dist = _dist()
    keys, counts = local_histogram(values)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return keys.view(np.float32), counts
    dev = _device()
    world = dist.get_world_size()
    n = torch.tensor([len(keys)], dtype=torch.int64, device=dev)
    ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ns, n)
    cap = int(max(int(x.item()) for x in ns))
    pad = torch.full((max(cap, 1),), -1, dtype=torch.int64, device=dev)
    pad[:len(keys)] = torch.from_numpy(keys.astype(np.int64)).to(dev)
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad)
    allk = torch.cat(gathered).cpu().numpy()
    union = np.unique(allk[allk >= 0]).astype(np.uint32)
    merged = np.zeros(len(union), dtype=np.int64)
    merged[np.searchsorted(union, keys)] = counts
    t = torch.from_numpy(merged).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return union.view(np.float32), t.cpu().numpy()


def percentile_from_histogram(keys, counts, q):
    """percentile_linear_interp (thresholds.rs:17-38) on a histogram: only xs[floor(x)] and xs[ceil(x)] of the sorted
    multiset are needed; they are found by walking the cumulative counts, and the interpolation itself is done by the
    library (mkp_percentile) on those two values so the f32 arithmetic is the product's, not numpy's."""
    n = int(counts.sum())
    if n < 2:
        raise ValueError("not enough datapoints, got %d" % n)
    q32 = np.float32(q)
    if q32 > np.float32(1.0):
        raise ValueError("quantile greater than 1.0")
    cum = np.cumsum(counts)

    def at(rank):
        return np.float32(keys[int(np.searchsorted(cum, rank, side="right"))])
    if q32 == np.float32(1.0):
        return float(at(n - 1))
    lq = np.float32(n - 1) * q32
    left, right = int(np.floor(lq)), int(np.ceil(lq))
    # rebuild a 2..3 element sorted array whose percentile at the same fractional rank equals the full one:
    # mkp_percentile(xs, n, q) reads xs[left] and xs[right] only, so hand it a sparse view through an index shift
    xs = (ctypes.c_float * 2)(float(at(left)), float(at(right)))
    frac = np.float32(lq - np.float32(np.trunc(lq)))
    out = ctypes.c_float()
    L = lib()
    L.mkp_percentile.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_uint64, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
    if left == right:
        return float(at(left))
    # two points, rank 1*frac: l = 1, lq' = frac, floor 0, ceil 1 -> xs[0]*(1-frac) + xs[1]*frac, the same f32 expression
    rc = L.mkp_percentile(xs, 2, ctypes.c_float(float(frac)), ctypes.byref(out))
    if rc != 0:
        raise ValueError("mkp_percentile failed (%d)" % rc)
    return float(out.value)


def estimate_thresholds_allreduce(ctx, bam, flags=(), device=None):
    """Per-base pass thresholds over the samples of ALL ranks (each rank samples its own BAM / windows).
    Interim: rank 0 estimates on its data and broadcasts; replaced by the histogram all-reduce once the device-side
    histogram entry points exist."""
    dist = _dist()
    thr = ctx.estimate_thresholds(bam, [f for f in flags if f != "--cpg"][:0]) if (not dist.is_initialized() or dist.get_rank() == 0) else None
    return broadcast_thresholds(thr, src=0)


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
