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
LIB_PATH = os.path.join(CSRC, "libmkpileup.so")

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
        L.mkp_last_error.restype = ctypes.c_char_p
        L.mkp_last_error.argtypes = [ctypes.c_void_p]
        L.mkp_pileup_main.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p, ctypes.c_size_t]
        L.mkp_ctx_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.mkp_ctx_destroy.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


EXPORTS = ["mkp_ctx_create", "mkp_ctx_destroy", "mkp_last_error", "mkp_version", "mkp_set_caller", "mkp_shard_begin",
           "mkp_shard_add_records", "mkp_shard_run", "mkp_shard_rerun", "mkp_get_stats", "mkp_process_region", "mkp_pileup_main",
           "mkp_percentile"]


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
