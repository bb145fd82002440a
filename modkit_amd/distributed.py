"""Multi-GPU glue for libmkpileup: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The pileup path itself shards by reference windows with no data-path collective
(mkp_pileup_main --gpus-rank R --gpus-world W); the only thing ranks ever share is the pass threshold:

* broadcast_thresholds  — default sampled mode (`-n 10042` / `-f x`): rank 0 estimates (mkp_estimate_thresholds), everyone receives it.
* estimate_thresholds_allreduce / percentile_from_histograms — full-data percentile (`-f 1.0`, thresholds.rs:121-159) when
  every rank has decoded only its own windows: the sample stays in each GPU's HBM, two-level histograms of the f32 *bit
  patterns* (top 16 bits, then the low 16 bits inside the bins that hold the wanted order statistics) are summed across ranks
  (int64 all-reduce, 512 KiB each) and each rank evaluates percentile_linear_interp (thresholds.rs:17-38) — bit-identical to
  sorting the union of all values.
* pileup_sharded — the launcher: thresholds as above, then `mkp_pileup_main --gpus-rank R --gpus-world W` per rank and an
  ordered concatenation of the rank outputs.
"""
import ctypes
import os

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


class RcclComm:
    """An RCCL communicator of this job's ranks (one GPU each), created through librccl's C API — ncclGetUniqueId on rank 0, the id
    handed round with torch.distributed, ncclCommInitRank — so that the C ABI's mkp_histogram_allreduce can run ncclAllReduce on the
    histograms in HBM.  (torch's own NCCL communicator is not reachable from outside; a Rust host would make the same three calls.)"""

    def __init__(self, rank=None, world=None, device=None):
        import torch
        dist = _dist()
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.lib = None
        # The librccl that sits next to the HIP runtime libmkpileup is bound to: the histograms are device pointers of THAT runtime, and
        # a process that imports torch may hold a second copy of the ROCm libraries (torch ships its own) whose HSA runtime nobody has
        # initialised ("no ROCm-capable device is detected" from ncclCommInitRank).
        names = []
        if os.environ.get("MKP_RCCL_LIB"):
            names.append(os.environ["MKP_RCCL_LIB"])
        try:
            L = lib()
            L.mkp_internal_hip_runtime_path.restype = ctypes.c_char_p
            hip_path = (L.mkp_internal_hip_runtime_path() or b"").decode()
            if hip_path:
                for cand in ("librccl.so.1", "librccl.so"):
                    full = os.path.join(os.path.dirname(os.path.realpath(hip_path)), cand)
                    if os.path.exists(full) and full not in names:
                        names.append(full)
        except (OSError, AttributeError):
            pass
        names += ["librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"]
        for name in names:
            try:
                self.lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                self.lib_path = name
                break
            except OSError:
                continue
        if self.lib is None:
            raise OSError("librccl.so not found")
        os.environ["MKP_RCCL_LIB"] = self.lib_path   # mkp_histogram_allreduce must call into the same copy the communicator lives in
        uid = (ctypes.c_char * 128)()
        if self.rank == 0 and self.lib.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        if self.world > 1:
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone().to(_device())
            dist.broadcast(t, src=0)
            ctypes.memmove(uid, bytes(t.cpu().numpy().tobytes()), 128)
        if device is not None:
            torch.cuda.set_device(device)

        class _Uid(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_char * 128)]
        u = _Uid()
        ctypes.memmove(ctypes.byref(u), uid, 128)
        self.handle = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _Uid, ctypes.c_int]
        rc = self.lib.ncclCommInitRank(ctypes.byref(self.handle), self.world, u, self.rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed (%d)" % rc)

    def close(self):
        if self.lib is not None and self.handle:
            self.lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self.lib.ncclCommDestroy(self.handle)
            self.handle = ctypes.c_void_p()


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


def percentile_from_histograms(get, q, reduced=False):
    """percentile_linear_interp (thresholds.rs:17-38) of the union of all ranks' samples of one base.
    get(level, prefix) -> this rank's uint64[65536] histogram (Context.histogram_get or mkp_histogram_from_values);
    every call is followed by a SUM all-reduce, so all ranks must call this function in the same order.
    reduced=True: get already returns the sum over the ranks (Context.histogram_allreduce: RCCL on the device buffers).
    Returns (value or None when the union is empty, n)."""
    L = lib()
    _red = (lambda h: h) if reduced else _allreduce_u64
    h0 = _red(np.ascontiguousarray(get(0, 0), dtype=np.uint64))
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
            h1 = _red(np.ascontiguousarray(get(1, int(bins[k])), dtype=np.uint64))
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


def estimate_thresholds_allreduce(ctx, bam, flags=(), q=0.1, rank=None, world=None, rccl_direct=None):
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
    # rccl_direct (or MKP_RCCL_DIRECT=1; nccl backend, one GPU per rank): the sum runs where the histograms sit — ncclAllReduce(u64) through
    # the C ABI (mkp_histogram_allreduce) on a communicator of this job's ranks.  Default: mkp_histogram_get + torch.distributed's
    # all_reduce, the path every multi-rank test here exercises (gloo: several ranks share the box's one GPU, which RCCL refuses); the
    # direct path has run on one rank only (tests/test_gpu_scale.py) until a multi-GPU node confirms it.
    comm = None
    if rccl_direct is None:
        rccl_direct = _env_flag("MKP_RCCL_DIRECT")
    if rccl_direct and dist.is_initialized() and dist.get_backend() == "nccl":
        comm = RcclComm()
    try:
        for b in BASES:
            if comm is not None:
                t, n = percentile_from_histograms(lambda level, prefix, b=b: ctx.histogram_allreduce(comm, b, level, prefix), q, reduced=True)
            else:
                t, n = percentile_from_histograms(lambda level, prefix, b=b: ctx.histogram_get(b, level, prefix), q)
            if t is not None:
                out[b] = t
    finally:
        if comm is not None:
            comm.close()
    return out


def _env_flag(name):
    import os
    return os.environ.get(name, "") == "1"


SAMPLING_FLAGS_WITH_VALUE = ("--region", "--sample-region", "--include-bed", "--include-positions", "--edge-filter", "--ignore", "--preset",
                             "--sampling-interval-size", "-t", "--threads")
SAMPLING_FLAGS_BARE = ("--include-unmapped", "--invert-edge-filter")


ESTIMATE_FLAGS_WITH_VALUE = ("-n", "--num-reads", "-f", "--sampling-frac", "-p", "--filter-percentile")


def split_threshold_flags(flags, q=None, mode=None):
    """`modkit pileup` flags -> (sampling flags [region, BED, collapse, edge filter ...], estimate flags [-n -f -p], the flags the sharded
    runs keep [everything but the estimate flags], threshold mode, percentile).  Mode as the subcommand itself decides: thresholds given
    (`--filter-threshold` / `--no-filtering`), full-data percentile (`-f 1.0`), else the count-based sampled estimate."""
    sflags, eflags, rest, i = [], [], [], 0
    frac, given = None, False
    flags = list(flags)
    while i < len(flags):
        f = flags[i]
        if f in SAMPLING_FLAGS_WITH_VALUE and i + 1 < len(flags):
            sflags += flags[i:i + 2]; rest += flags[i:i + 2]; i += 2
        elif f in ESTIMATE_FLAGS_WITH_VALUE and i + 1 < len(flags):
            eflags += flags[i:i + 2]
            if f in ("-f", "--sampling-frac"):
                frac = float(flags[i + 1])
            if f in ("-p", "--filter-percentile") and q is None:
                q = float(flags[i + 1])
            i += 2
        else:
            if f in SAMPLING_FLAGS_BARE:
                sflags.append(f)
            if f in ("--filter-threshold", "--no-filtering"):
                given = True
            rest.append(f); i += 1
    if q is None:
        q = 0.1
    if mode is None:
        mode = "given" if given else "full" if (frac is not None and frac >= 1.0) else "sampled"
    if mode not in ("sampled", "full", "given"):
        raise ValueError("pileup_sharded: mode must be 'sampled', 'full' or 'given'")
    return sflags, eflags, rest, mode, q


def reduce_thresholds(ctx, q, rccl_direct=None):
    """Per-base pass thresholds from the histograms every rank's context holds (mkp_histogram_*): summed over the ranks — RCCL on the
    device buffers behind the C ABI (rccl_direct / MKP_RCCL_DIRECT=1, nccl backend), else mkp_histogram_get + torch.distributed — then the
    exact percentile of the union.  All ranks call this in step."""
    dist = _dist()
    comm = None
    if rccl_direct is None:
        rccl_direct = _env_flag("MKP_RCCL_DIRECT")
    if rccl_direct and dist.is_initialized() and dist.get_backend() == "nccl":
        comm = RcclComm()
    out = {}
    try:
        for b in BASES:
            if comm is not None:
                t, n = percentile_from_histograms(lambda level, prefix, b=b: ctx.histogram_allreduce(comm, b, level, prefix), q, reduced=True)
            else:
                t, n = percentile_from_histograms(lambda level, prefix, b=b: ctx.histogram_get(b, level, prefix), q)
            if t is not None:
                out[b] = t
    finally:
        if comm is not None:
            comm.close()
    return out


def pileup_sharded(argv, rank=None, world=None, device=None, q=None, stats=None, mode=None):
    """`modkit pileup` with ONE BAM sharded over the ranks of the current torch.distributed job (one process per GPU): every rank runs
    its contiguous run of the reference's interval grid into `<out>.rank<R>` and rank 0 concatenates the parts in rank order into `<out>`.
    argv = [in.bam, out.bed, flags...].  Each rank makes ONE call into the library (mkp_pileup_run_cb): its shards are ingested on the
    device from the first moment — several in flight, kept in HBM under a budget — and the pass thresholds arrive through a callback.
    Threshold modes (`mode`, default: by the flags, as `modkit pileup` itself decides):
      "sampled" — the reference's default: the count-based schedule (`-n 10042`, or `-f x < 1`) carries quotas from interval to interval
                  and does not shard; it reads only interval heads, so rank 0 estimates (mkp_estimate_thresholds on a context of its own,
                  beside its ingest) and broadcasts four floats, while every rank's shards are already on their way into HBM.
                  The output is byte-identical to a single-GPU run with the same flags.
      "full"    — `-f 1.0` (thresholds.rs:121-159): every rank samples the shards it has just ingested, FROM HBM (each read belongs to the
                  shard that holds its start, so the union over ranks is the single-rank sample), the two-level histograms are summed
                  over the ranks (the path's one collective) and every rank evaluates the percentile of the union; then the pileup pass
                  runs on the same resident shards — every block of the file is read, uploaded and inflated once, by one rank.
                  Byte-identical to a single-GPU run with `-f 1.0`.
      "given"   — `--filter-threshold` / `--no-filtering` in argv: nothing to estimate.
    Returns the thresholds used ({base: f32}; empty for "given").
    `--with-header` is written by rank 0 only; `--bgzf` (one BGZF stream + one index) and `--partition-tag` (one file per key)
    do not concatenate and are refused.  `stats` (a dict) receives this rank's wall times."""
    import os
    import shutil
    import time
    from . import Context
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
    sflags, eflags, rest, mode, q = split_threshold_flags(flags, q, mode)
    t0 = time.time()
    used, t_thr = {}, [0.0]

    def thresholds(have_sample):
        t1 = time.time()
        if have_sample:   # full-data mode: this rank's sample is in ctx's histograms
            thr = reduce_thresholds(ctx, q)
        else:             # count-based estimate: one rank's job
            thr = {}
            if rank == 0:
                c2 = Context(device=0 if device is None else device)
                try:
                    thr = c2.estimate_thresholds(bam, sflags + eflags)
                finally:
                    c2.close()
            thr = broadcast_thresholds(thr, src=0)
        if not thr:
            raise ValueError("no mod calls sampled on any rank")
        used.update(thr)
        t_thr[0] = time.time() - t1
        return thr

    part = "%s.rank%d" % (out, rank)
    run_flags = rest + (["-f", "1.0", "-p", repr(float(q))] if mode == "full" else []) + ["--gpus-rank", str(rank), "--gpus-world", str(world)]
    ctx = Context(device=0 if device is None else device)
    try:
        rep = ctx.pileup_run_cb([bam, part] + run_flags, thresholds)
    finally:
        ctx.close()
    if stats is not None:
        stats.update({"threshold_s": t_thr[0], "pileup_s": time.time() - t0 - t_thr[0], "total_s": time.time() - t0, "part_bytes": os.path.getsize(part),
                      "report": rep.as_dict()})
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
    return dict(used)


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
