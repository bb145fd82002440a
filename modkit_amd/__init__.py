"""modkit_amd — Python host mirror of libmkpileup (the MI355X implementation of `modkit pileup`).

The product is the C-ABI shared library built from modkit_amd/csrc (include/mkpileup.h); this module
only loads it and mirrors the reference's `modkit pileup` command surface:

    modkit_amd.pileup(["in.bam", "out.bed", "--cpg", "--ref", "ref.fa", ...])   # == `modkit pileup ...`

There is no CPU path: every call runs the HIP kernels and fails loudly when the library or a gfx950
device is missing.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MKP_LIB_PATH") or os.path.join(CSRC, "libmkpileup.so")   # MKP_LIB_PATH: another build of the same library (e.g. -DMKP_DEBUG ablations)

MKP_OK = 0
STATUS = {0: "MKP_OK", -1: "MKP_E_INVALID", -2: "MKP_E_IO", -3: "MKP_E_UNSUPPORTED", -4: "MKP_E_DEVICE", -5: "MKP_E_NOMEM",
          -6: "MKP_E_THRESHOLD"}


class MkpError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS.get(status, status), message))
        self.status = status


def build(force=False):
    """Compile libmkpileup.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "libmkpileup.so", "mkpileup"])
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MkpError(-4, "libmkpileup.so is not built (run modkit_amd.build() / __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        L.mkp_version.restype = ctypes.c_char_p
        L.mkp_host_threads.restype = ctypes.c_uint
        L.mkp_last_error.restype = ctypes.c_char_p
        L.mkp_last_error.argtypes = [ctypes.c_void_p]
        L.mkp_pileup_main.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p, ctypes.c_size_t]
        L.mkp_ctx_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.mkp_ctx_destroy.argtypes = [ctypes.c_void_p]
        L.mkp_set_caller.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mkp_estimate_thresholds.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p, ctypes.c_void_p]
        L.mkp_process_region.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mkp_shard_rerun.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.mkp_get_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mkp_pileup_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p]
        L.mkp_pileup_run_cb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), THRESHOLD_FN, ctypes.c_void_p, ctypes.c_void_p]
        L.mkp_pileup_hemi_main.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p, ctypes.c_size_t]
        L.mkp_extract_calls_main.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p, ctypes.c_size_t]
        L.mkp_pileup_hemi_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p]
        L.mkp_summary.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p]
        L.mkp_bgzf_inflate.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]
        L.mkp_hemi_shard_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_void_p]
        u64p, f32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float)
        L.mkp_sample_probs.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), f32p, ctypes.c_uint32, f32p, ctypes.c_void_p, u64p]
        L.mkp_histogram_begin.argtypes = [ctypes.c_void_p]
        L.mkp_histogram_add_bam.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
        L.mkp_histogram_get.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u64p]
        L.mkp_histogram_allreduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u64p]
        L.mkp_histogram_from_values.argtypes = [f32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, u64p]
        L.mkp_histogram_locate.argtypes = [u64p, ctypes.c_float, ctypes.POINTER(ctypes.c_uint32), u64p, u64p]
        L.mkp_histogram_resolve.argtypes = [ctypes.c_uint32, u64p, ctypes.c_uint64, f32p]
        L.mkp_percentile_from_histogram.argtypes = [ctypes.c_uint64, ctypes.c_float, ctypes.c_float, ctypes.c_float, f32p]
        # the structs this binding allocates mirror ONE revision of include/mkpileup.h: a library of another revision would write past them
        if os.environ.get("MKP_LIB_PATH") and not hasattr(L, "mkp_abi_version"):
            _lib = L   # (A/B runs against a build from before the query existed, tools/dbg/ab.sh)
            return _lib
        L.mkp_abi_version.restype = ctypes.c_uint32
        L.mkp_run_report_size.restype = ctypes.c_size_t
        if L.mkp_abi_version() != ABI_VERSION or L.mkp_run_report_size() != ctypes.sizeof(RunReport):
            raise MkpError(-1, "libmkpileup.so has ABI revision %d (mkp_run_report %d bytes), this binding %d (%d bytes): rebuild" %
                           (L.mkp_abi_version(), L.mkp_run_report_size(), ABI_VERSION, ctypes.sizeof(RunReport)))
        _lib = L
    return _lib


ABI_VERSION = 3   # MKP_ABI_VERSION of include/mkpileup.h


# mkp_threshold_fn (include/mkpileup.h): int fn(void* user, mkp_ctx* ctx, int have_sample, float thresholds[4], uint8_t has[4])
THRESHOLD_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint8))

EXPORTS = ["mkp_ctx_create", "mkp_ctx_destroy", "mkp_last_error", "mkp_version", "mkp_abi_version", "mkp_run_report_size", "mkp_host_threads", "mkp_set_caller", "mkp_shard_begin",
           "mkp_shard_add_records", "mkp_shard_set_intervals", "mkp_shard_run", "mkp_batch_run", "mkp_shard_rerun", "mkp_get_stats", "mkp_process_region", "mkp_pileup_main",
           "mkp_pileup_run", "mkp_pileup_run_cb", "mkp_percentile", "mkp_estimate_thresholds", "mkp_host_mm_ranks", "mkp_host_map_order",
           "mkp_set_partition_tags", "mkp_histogram_begin", "mkp_histogram_add_bam", "mkp_histogram_get", "mkp_histogram_allreduce", "mkp_histogram_from_values", "mkp_histogram_locate",
           "mkp_histogram_resolve", "mkp_percentile_from_histogram", "mkp_hemi_shard_run", "mkp_pileup_hemi_main", "mkp_pileup_hemi_run", "mkp_bgzf_inflate", "mkp_sample_probs", "mkp_summary", "mkp_extract_calls_main"]


def pileup(argv):
    """`modkit pileup` (ModBamPileup::run, src/pileup/subcommand.rs:382): argv = [in_bam, out_bed, flags...]."""
    L = lib()
    args = [str(a).encode() for a in argv]
    arr = (ctypes.c_char_p * len(args))(*args)
    err = ctypes.create_string_buffer(2048)
    rc = L.mkp_pileup_main(len(args), arr, err, len(err))
    if rc != MKP_OK:
        raise MkpError(rc, err.value.decode(errors="replace"))
    return rc


def pileup_hemi(argv):
    """`modkit pileup-hemi` (DuplexModBamPileup::run, src/pileup/subcommand.rs:1122): argv = [in_bam, "-o", out_bed, flags...]."""
    L = lib()
    args = [str(a).encode() for a in argv]
    arr = (ctypes.c_char_p * len(args))(*args)
    err = ctypes.create_string_buffer(2048)
    rc = L.mkp_pileup_hemi_main(len(args), arr, err, len(err))
    if rc != MKP_OK:
        raise MkpError(rc, err.value.decode(errors="replace"))
    return rc


def extract_calls(argv):
    """`modkit extract calls` (EntryExtractCalls::run, src/extract/subcommand.rs:452): argv = [in_bam, out_tsv, flags...]."""
    L = lib()
    args = [str(a).encode() for a in argv]
    arr = (ctypes.c_char_p * len(args))(*args)
    err = ctypes.create_string_buffer(2048)
    rc = L.mkp_extract_calls_main(len(args), arr, err, len(err))
    if rc != MKP_OK:
        raise MkpError(rc, err.value.decode(errors="replace"))
    return rc


# ---------------------------------------------------------------------------------------------
# ctypes mirror of include/mkpileup.h for callers that drive shards directly (bench.py, tests)
class Config(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("tile_positions", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 6)]


class ModThreshold(ctypes.Structure):
    _fields_ = [("code_repr", ctypes.c_uint32), ("threshold", ctypes.c_float)]


class Caller(ctypes.Structure):
    _fields_ = [("default_threshold", ctypes.c_float), ("per_base_threshold", ctypes.c_float * 4), ("has_per_base", ctypes.c_uint8 * 4),
                ("per_mod", ctypes.POINTER(ModThreshold)), ("n_per_mod", ctypes.c_uint32), ("numeric_mode", ctypes.c_uint32),
                ("collapse_code", ctypes.c_uint32), ("edge_filter", ctypes.c_uint32), ("edge_start", ctypes.c_uint32),
                ("edge_end", ctypes.c_uint32), ("edge_inverted", ctypes.c_uint32), ("force_allow_implicit", ctypes.c_uint32),
                ("combine_strands", ctypes.c_uint32), ("max_depth", ctypes.c_uint32)]


class Shard(ctypes.Structure):
    _fields_ = [("tid", ctypes.c_int32), ("start", ctypes.c_uint32), ("end", ctypes.c_uint32), ("focus", ctypes.c_void_p),
                ("combos", ctypes.c_void_p), ("n_combos", ctypes.c_uint32)]


class Rows(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_uint64), ("pos", ctypes.POINTER(ctypes.c_uint32)), ("strand", ctypes.POINTER(ctypes.c_uint8)),
                ("code_repr", ctypes.POINTER(ctypes.c_uint32)), ("motif_idx", ctypes.POINTER(ctypes.c_int32)),
                ("n_valid", ctypes.POINTER(ctypes.c_uint32)), ("n_mod", ctypes.POINTER(ctypes.c_uint32)),
                ("n_canonical", ctypes.POINTER(ctypes.c_uint32)), ("n_other", ctypes.POINTER(ctypes.c_uint32)),
                ("n_delete", ctypes.POINTER(ctypes.c_uint32)), ("n_fail", ctypes.POINTER(ctypes.c_uint32)),
                ("n_diff", ctypes.POINTER(ctypes.c_uint32)), ("n_nocall", ctypes.POINTER(ctypes.c_uint32)),
                ("processed_records", ctypes.c_uint64), ("skipped_records", ctypes.c_uint64),
                ("partition_key", ctypes.POINTER(ctypes.c_uint32)), ("n_partition_keys", ctypes.c_uint32),
                ("partition_key_names", ctypes.POINTER(ctypes.c_char_p))]


class HemiRows(ctypes.Structure):
    """mkp_hemi_rows: duplex pattern counts (DuplexPatternCounts, src/pileup/duplex.rs:32-56)."""
    _u32p = ctypes.POINTER(ctypes.c_uint32)
    _fields_ = [("n_rows", ctypes.c_uint64), ("pos", _u32p), ("primary_base", ctypes.POINTER(ctypes.c_uint8)), ("pattern_pos", _u32p),
                ("pattern_neg", _u32p), ("n_valid", _u32p), ("count", _u32p), ("n_canonical", _u32p), ("n_other_pattern", _u32p),
                ("n_delete", _u32p), ("n_fail", _u32p), ("n_diff", _u32p), ("n_nocall", _u32p),
                ("processed_records", ctypes.c_uint64), ("skipped_records", ctypes.c_uint64)]


HEMI_ROW_FIELDS = ("pos", "primary_base", "pattern_pos", "pattern_neg", "n_valid", "count", "n_canonical", "n_other_pattern", "n_delete",
                   "n_fail", "n_diff", "n_nocall")


class SummaryOut(ctypes.Structure):
    """mkp_summary_out: ModSummary (src/summarize.rs:21-46) as counts."""
    _fields_ = [("total_reads_used", ctypes.c_uint64), ("reads_with_mod_calls", ctypes.c_uint64 * 4), ("threshold", ctypes.c_float * 4),
                ("has_threshold", ctypes.c_uint8 * 4), ("n_rows", ctypes.c_uint32), ("base", ctypes.POINTER(ctypes.c_uint8)),
                ("code_repr", ctypes.POINTER(ctypes.c_uint32)), ("pass_count", ctypes.POINTER(ctypes.c_uint64)), ("fail_count", ctypes.POINTER(ctypes.c_uint64))]


class Stats(ctypes.Structure):
    _fields_ = [("pack_ms", ctypes.c_double), ("h2d_ms", ctypes.c_double), ("kernel_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double),
                ("decode_kernel_ms", ctypes.c_double), ("pileup_kernel_ms", ctypes.c_double), ("gather_kernel_ms", ctypes.c_double),
                ("n_reads", ctypes.c_uint64), ("n_events", ctypes.c_uint64), ("n_rows", ctypes.c_uint64), ("n_tiles", ctypes.c_uint64),
                ("n_positions", ctypes.c_uint64), ("alg_bytes_decode", ctypes.c_uint64), ("alg_bytes_pileup", ctypes.c_uint64),
                ("rows_kernel_ms", ctypes.c_double), ("alg_bytes_rows", ctypes.c_uint64),
                ("stream_bytes", ctypes.c_uint64), ("alg_bytes_agg_survey", ctypes.c_uint64), ("slot_pipeline", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class RunReport(ctypes.Structure):
    _fields_ = [("load_ms", ctypes.c_double), ("threshold_ms", ctypes.c_double), ("focus_ms", ctypes.c_double), ("pack_ms", ctypes.c_double),
                ("h2d_ms", ctypes.c_double), ("kernel_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double), ("write_ms", ctypes.c_double),
                ("total_ms", ctypes.c_double), ("n_rows", ctypes.c_uint64), ("n_positions", ctypes.c_uint64), ("n_shards", ctypes.c_uint64),
                ("processed_records", ctypes.c_uint64), ("skipped_records", ctypes.c_uint64), ("threshold", ctypes.c_float * 4),
                ("has_threshold", ctypes.c_uint8 * 4),
                ("grid_wait_ms", ctypes.c_double), ("callback_ms", ctypes.c_double), ("ingest_kernel_ms", ctypes.c_double), ("ingest_upload_ms", ctypes.c_double),
                ("ingest_table_ms", ctypes.c_double), ("ingest_pack_ms", ctypes.c_double), ("ingest_comp_bytes", ctypes.c_uint64), ("ingest_raw_bytes", ctypes.c_uint64),
                ("ingest_blocks", ctypes.c_uint64), ("ingest_records", ctypes.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("threshold", "has_threshold")}
        d["thresholds"] = {"ACGT"[i]: float(self.threshold[i]) for i in range(4) if self.has_threshold[i]}
        return d


class Context:
    """One mkp_ctx (one per host thread; binds to one GPU)."""

    def __init__(self, device=0, tile_positions=0):
        self.L = lib()
        self.h = ctypes.c_void_p()
        cfg = Config(device=device, tile_positions=tile_positions)
        rc = self.L.mkp_ctx_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != MKP_OK:
            raise MkpError(rc, "mkp_ctx_create failed (no gfx950 device visible?)")

    def _check(self, rc):
        if rc != MKP_OK:
            raise MkpError(rc, self.L.mkp_last_error(self.h).decode(errors="replace"))

    def close(self):
        if self.h:
            self.L.mkp_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def set_caller(self, default_threshold=0.0, per_base=None, per_mod=None, numeric_mode=0, collapse_code=0, combine_strands=False,
                   force_allow_implicit=False, edge=None, max_depth=8000):
        c = Caller()
        c.default_threshold = default_threshold
        for b, t in (per_base or {}).items():
            i = "ACGT".index(b) if isinstance(b, str) else int(b)
            c.per_base_threshold[i] = t
            c.has_per_base[i] = 1
        pm = list((per_mod or {}).items())
        arr = (ModThreshold * max(1, len(pm)))()
        for i, (code, t) in enumerate(pm):
            arr[i].code_repr = ord(code) if isinstance(code, str) else int(code)
            arr[i].threshold = t
        c.per_mod = arr
        c.n_per_mod = len(pm)
        c.numeric_mode, c.collapse_code, c.combine_strands = numeric_mode, collapse_code, int(combine_strands)
        c.force_allow_implicit, c.max_depth = int(force_allow_implicit), max_depth
        if edge:
            c.edge_filter, c.edge_start, c.edge_end, c.edge_inverted = 1, edge[0], edge[1], int(edge[2]) if len(edge) > 2 else 0
        self._check(self.L.mkp_set_caller(self.h, ctypes.byref(c)))

    def estimate_thresholds(self, bam, argv=()):
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * max(1, len(args)))(*args)
        thr = (ctypes.c_float * 4)()
        has = (ctypes.c_uint8 * 4)()
        self._check(self.L.mkp_estimate_thresholds(self.h, str(bam).encode(), len(args), arr, thr, has))
        return {"ACGT"[i]: float(thr[i]) for i in range(4) if has[i]}

    def histogram_begin(self):
        self._check(self.L.mkp_histogram_begin(self.h))

    def histogram_add_bam(self, bam, argv=()):
        """Sample this rank's windows of `bam` (argv: sampling flags, --gpus-rank/--gpus-world) into the HBM-resident sample."""
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * max(1, len(args)))(*args)
        self._check(self.L.mkp_histogram_add_bam(self.h, str(bam).encode(), len(args), arr))

    def histogram_allreduce(self, comm, base, level, prefix=0):
        """histogram_get summed over the ranks of an RCCL communicator (modkit_amd.distributed.RcclComm), in HBM over xGMI."""
        import numpy as np
        out = np.zeros(65536, dtype=np.uint64)
        i = "ACGT".index(base) if isinstance(base, str) else int(base)
        self._check(self.L.mkp_histogram_allreduce(self.h, comm.handle, i, level, prefix, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out

    def histogram_get(self, base, level, prefix=0):
        """uint64[65536] numpy array: level 0 = top 16 bits of the f32 patterns of `base`'s sample, level 1 = low 16 bits under `prefix`."""
        import numpy as np
        out = np.zeros(65536, dtype=np.uint64)
        i = "ACGT".index(base) if isinstance(base, str) else int(base)
        self._check(self.L.mkp_histogram_get(self.h, i, level, prefix, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out

    def process_region(self, bam, tid, start, end):
        sh = Shard(tid=tid, start=start, end=end, focus=None, combos=None, n_combos=0)
        rows = Rows()
        self._check(self.L.mkp_process_region(self.h, str(bam).encode(), ctypes.byref(sh), ctypes.byref(rows)))
        return rows

    def pileup_run(self, argv):
        """`modkit pileup` on this context (mkp_pileup_run): the last shard stays resident for rerun(); returns the stage report."""
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * len(args))(*args)
        rep = RunReport()
        self._check(self.L.mkp_pileup_run(self.h, len(args), arr, ctypes.byref(rep)))
        return rep

    def pileup_run_cb(self, argv, thresholds):
        """`modkit pileup` with the pass thresholds decided by the caller (mkp_pileup_run_cb — the multi-GPU form: argv carries --gpus-rank /
        --gpus-world and no threshold flags).  The library ingests this rank's shards ahead, then calls `thresholds(have_sample)` once:
        have_sample True (`-f 1.0`): this context's histograms hold the sample of this rank's shards — reduce them over the ranks and
        evaluate the percentile; False: nothing was sampled, supply the values.  It returns {base letter: f32}.  Returns the stage report."""
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * len(args))(*args)
        rep = RunReport()
        failure = []

        def _cb(user, ctx, have_sample, thr, has):
            try:
                got = thresholds(bool(have_sample))
                for i, b in enumerate("ACGT"):
                    if b in got:
                        thr[i] = float(got[b]); has[i] = 1
                    else:
                        thr[i] = 0.0; has[i] = 0
                return MKP_OK
            except BaseException as e:   # (must not unwind through the C frames)
                failure.append(e)
                return -2
        fn = THRESHOLD_FN(_cb)
        rc = self.L.mkp_pileup_run_cb(self.h, len(args), arr, fn, None, ctypes.byref(rep))
        if failure:
            raise failure[0]
        self._check(rc)
        return rep

    def pileup_hemi_run(self, argv):
        """`modkit pileup-hemi` on this context (mkp_pileup_hemi_run); returns the stage report."""
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * len(args))(*args)
        rep = RunReport()
        self._check(self.L.mkp_pileup_hemi_run(self.h, len(args), arr, ctypes.byref(rep)))
        return rep

    def hemi_shard_run(self, partner_offset, interval_starts=()):
        """mkp_hemi_shard_run on the shard begun with shard_begin / add_records; returns a dict of numpy arrays."""
        import numpy as np
        iv = (ctypes.c_uint32 * max(1, len(interval_starts)))(*[int(x) for x in interval_starts])
        rows = HemiRows()
        self._check(self.L.mkp_hemi_shard_run(self.h, int(partner_offset), iv if len(interval_starts) else None, len(interval_starts), ctypes.byref(rows)))
        n = int(rows.n_rows)
        out = {f: (np.ctypeslib.as_array(getattr(rows, f), shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint32)) for f in HEMI_ROW_FIELDS}
        out["processed_records"], out["skipped_records"] = int(rows.processed_records), int(rows.skipped_records)
        return out

    def sample_probs(self, bam, percentiles=(0.1, 0.5, 0.9), argv=()):
        """`modkit sample-probs` percentiles (mkp_sample_probs): {base: {"n": sampled calls, "percentiles": {q: value}}}; values are f32."""
        import numpy as np
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * max(1, len(args)))(*args)
        qs = np.asarray(percentiles, dtype=np.float32)
        vals = np.zeros((4, len(qs)), dtype=np.float32)
        has = (ctypes.c_uint8 * 4)()
        n = (ctypes.c_uint64 * 4)()
        f32p = ctypes.POINTER(ctypes.c_float)
        self._check(self.L.mkp_sample_probs(self.h, str(bam).encode(), len(args), arr, qs.ctypes.data_as(f32p), len(qs), vals.ctypes.data_as(f32p), has, n))
        return {"ACGT"[b]: {"n": int(n[b]), "percentiles": {float(qs[k]): vals[b, k] for k in range(len(qs))}} for b in range(4) if has[b]}

    def summary(self, bam, argv=()):
        """`modkit summary` as counts (mkp_summary): {"total", "reads_with": {base: n}, "threshold": {base: f32}, "rows": {(base, code): (pass, fail)}};
        code "-" = canonical, a letter, or a ChEBI number as text."""
        args = [str(a).encode() for a in argv]
        arr = (ctypes.c_char_p * max(1, len(args)))(*args)
        o = SummaryOut()
        self._check(self.L.mkp_summary(self.h, str(bam).encode(), len(args), arr, ctypes.byref(o)))
        def code(c):
            return "-" if c == 0 else str(c & 0x7fffffff) if c & 0x80000000 else chr(c)
        return {"total": int(o.total_reads_used), "reads_with": {"ACGT"[b]: int(o.reads_with_mod_calls[b]) for b in range(4) if o.reads_with_mod_calls[b]},
                "threshold": {"ACGT"[b]: float(o.threshold[b]) for b in range(4) if o.has_threshold[b]},
                "rows": {("ACGT"[o.base[i]], code(o.code_repr[i])): (int(o.pass_count[i]), int(o.fail_count[i])) for i in range(o.n_rows)}}

    def bgzf_inflate(self, data):
        """mkp_bgzf_inflate: inflate a BGZF file image (bytes) on the device; returns (inflated bytes, kernel ms)."""
        out, n, ms = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_double()
        self._check(self.L.mkp_bgzf_inflate(self.h, data, len(data), ctypes.byref(out), ctypes.byref(n), ctypes.byref(ms)))
        return ctypes.string_at(out, n.value) if n.value else b"", ms.value

    def rerun(self, iters, fetch=False):
        rows = Rows()
        self._check(self.L.mkp_shard_rerun(self.h, int(iters), ctypes.byref(rows) if fetch else None))
        return rows

    def stats(self):
        s = Stats()
        self._check(self.L.mkp_get_stats(self.h, ctypes.byref(s)))
        return s


ROW_FIELDS = ("pos", "strand", "code_repr", "motif_idx", "n_valid", "n_mod", "n_canonical", "n_other", "n_delete", "n_fail", "n_diff", "n_nocall")


def read_bedmethyl(path):
    """The count columns of a bedMethyl file (writers.rs:87-156) as the row arrays mkp_rows carries: pos, strand, code_repr (a letter's
    code point, or the ChEBI number | 1 << 31), n_valid, n_mod, n_canonical, n_other, n_delete, n_fail, n_diff, n_nocall — so that rows
    fetched from the device can be compared with a file that was itself compared with the oracle's."""
    import numpy as np
    import pandas as pd
    cols = {1: "pos", 3: "code", 5: "strand", 9: "n_valid", 11: "n_mod", 12: "n_canonical", 13: "n_other", 14: "n_delete", 15: "n_fail", 16: "n_diff", 17: "n_nocall"}
    if os.path.getsize(path) == 0:
        return {f: np.zeros(0, dtype=np.uint32) for f in ROW_FIELDS if f != "motif_idx"}
    df = pd.read_csv(path, sep=r"\s+", header=None, usecols=sorted(cols), dtype={3: str, 5: str}, engine="c")
    out = {name: df[k].to_numpy().astype(np.uint32) for k, name in cols.items() if name not in ("code", "strand")}
    out["strand"] = df[5].map(ord).to_numpy().astype(np.uint8)
    out["code_repr"] = df[3].map(lambda c: (int(c) | (1 << 31)) if c.isdigit() else ord(c)).to_numpy().astype(np.uint32)
    return out


def rows_digest(r, fields=("pos", "strand", "code_repr", "n_valid", "n_mod", "n_canonical", "n_other", "n_delete", "n_fail", "n_diff", "n_nocall")):
    """sha256 over the row arrays (numpy dicts from rows_to_numpy / read_bedmethyl), field by field, widths normalised."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for f in fields:
        h.update(np.ascontiguousarray(np.asarray(r[f]).astype(np.uint32)).tobytes())
    return h.hexdigest()


def rows_to_numpy(rows):
    """Copy an mkp_rows view (owned by the ctx until its next run) into a dict of numpy arrays."""
    import numpy as np
    n = int(rows.n_rows)
    out = {}
    for f in ROW_FIELDS:
        p = getattr(rows, f)
        out[f] = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint32)
    return out
