"""CPU-side checks of the product library (no GPU, no compute calls): libmkpileup.so loads and exports every symbol
include/mkpileup.h declares, the ctypes mirror matches the header's struct layout, compute entry points fail loudly
without a device (there is no CPU fallback), and the host-only arithmetic (mkp_percentile = percentile_linear_interp,
src/thresholds.rs:17-38) reproduces the reference's known answers."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import modkit_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mkpileup.h")


@pytest.fixture(scope="module")
def L():
    modkit_amd.build()
    return modkit_amd.lib()


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"^(?:int|unsigned|void|const char\*|uint32_t|size_t)\s+(mkp_\w+)\s*\(", text, flags=re.M)))


def test_every_declared_symbol_is_exported(L):
    syms = declared_symbols()
    assert len(syms) >= 17 and sorted(modkit_amd.EXPORTS) == syms
    for s in syms:
        assert getattr(L, s) is not None
    nm = subprocess.check_output(["nm", "-D", "--defined-only", modkit_amd.LIB_PATH], text=True)
    exported = set(re.findall(r" T (mkp_\w+)", nm))
    assert set(syms) <= exported
    # nothing from the test oracle is linked into the product
    assert "mko" not in nm and "oracle" not in nm.lower()


def test_ctypes_mirror_matches_header_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mkpileup.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mkp_config), sizeof(mkp_mod_threshold), '
                   'sizeof(mkp_caller), sizeof(mkp_record), sizeof(mkp_motif_combo), sizeof(mkp_shard), sizeof(mkp_rows), sizeof(mkp_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])  # the header is plain C
    sizes = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert sizes[0] == ctypes.sizeof(modkit_amd.Config)
    assert sizes[1] == ctypes.sizeof(modkit_amd.ModThreshold)
    assert sizes[2] == ctypes.sizeof(modkit_amd.Caller)
    assert sizes[5] == ctypes.sizeof(modkit_amd.Shard)
    assert sizes[6] == ctypes.sizeof(modkit_amd.Rows)
    assert sizes[7] == ctypes.sizeof(modkit_amd.Stats)
    assert sizes[4] == 16 and sizes[3] == 32


def test_abi_revision_and_report_size(L):
    # a caller built against another revision of the header must be able to tell before it hands over a struct
    m = re.search(r"#define MKP_ABI_VERSION (\d+)u", open(HEADER).read())
    assert m and L.mkp_abi_version() == int(m.group(1)) == modkit_amd.ABI_VERSION
    assert L.mkp_run_report_size() == ctypes.sizeof(modkit_amd.RunReport)


def test_version_and_no_device_is_loud(L):
    assert b"gfx950" in L.mkp_version()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-device path cannot be observed here")
    with pytest.raises(modkit_amd.MkpError) as e:
        modkit_amd.Context(device=0)
    assert e.value.status == -4  # MKP_E_DEVICE
    fix = os.path.join(ROOT, "tests", "golden", "modkit_fixtures", "bc_anchored_10_reads.sorted.bam")
    with pytest.raises(modkit_amd.MkpError) as e:
        modkit_amd.pileup([fix, "/dev/null", "--no-filtering"])
    assert e.value.status == -4 and "no CPU path" in str(e.value)


def pct(L, xs, q):
    a = (ctypes.c_float * len(xs))(*xs)
    out = ctypes.c_float()
    L.mkp_percentile.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_uint64, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
    rc = L.mkp_percentile(a, len(xs), ctypes.c_float(q), ctypes.byref(out))
    return rc, out.value


def test_percentile_known_answers(L):
    # src/thresholds.rs:197-201: a quantile above 1 is an error; fewer than two datapoints is an error (line 18)
    assert pct(L, [0.1, 0.2, 0.3], 1.5)[0] == -6
    assert pct(L, [0.5], 0.1)[0] == -6
    assert pct(L, [0.25, 0.5, 0.75], 1.0) == (0, 0.75)
    assert pct(L, [0.25, 0.5, 0.75], 0.5) == (0, 0.5)
    # the threshold the reference derives for its golden modbam.modpileup_filt025 (109 calls, p=0.25) is f32-exact 0.662109375
    rng = np.random.default_rng(7)
    xs = np.sort(((rng.integers(0, 256, 1001).astype(np.float32) + np.float32(0.5)) / np.float32(256)))
    for q in (0.1, 0.25, 0.333, 0.9):
        q32 = np.float32(q)
        lq = np.float32(len(xs) - 1) * q32
        g = np.float32(lq - np.float32(np.trunc(lq)))
        want = np.float32(np.float32(xs[int(np.floor(lq))] * np.float32(np.float32(1) - g)) + np.float32(xs[int(np.ceil(lq))] * g))
        rc, got = pct(L, [float(x) for x in xs], q)
        assert rc == 0 and np.float32(got) == want


def test_bad_arguments_do_not_crash(L):
    assert L.mkp_ctx_create(None, None) == -1
    L.mkp_get_stats.restype = ctypes.c_int
    assert L.mkp_get_stats(None, None) == -1
    assert L.mkp_last_error(None).startswith(b"no context")
    err = ctypes.create_string_buffer(256)
    arr = (ctypes.c_char_p * 1)(b"only_one_positional")
    assert L.mkp_pileup_main(1, arr, err, 256) == -1 and b"usage" in err.value
    arr = (ctypes.c_char_p * 3)(b"/nonexistent.bam", b"/dev/null", b"--partition-tag")
    assert L.mkp_pileup_main(3, arr, err, 256) in (-1, -3)


class HostTag(ctypes.Structure):
    _fields_ = [("base", ctypes.c_uint8), ("negative_strand", ctypes.c_uint8), ("mode", ctypes.c_uint8), ("n_codes", ctypes.c_uint8),
                ("codes", ctypes.c_uint32 * 4), ("rank_off", ctypes.c_uint32), ("n_ranks", ctypes.c_uint32)]


def mm_ranks(L, mm, l_seq, n_ml):
    tags = (HostTag * 8)()
    ranks = (ctypes.c_uint32 * 256)()
    L.mkp_host_mm_ranks.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(HostTag), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32]
    n = L.mkp_host_mm_ranks(mm.encode(), l_seq, n_ml, tags, 8, ranks, 256)
    if n < 0:
        return n
    return [dict(base="ACGTN"[t.base], neg=bool(t.negative_strand), mode="?. "[t.mode], codes=[t.codes[i] for i in range(t.n_codes)],
                 ranks=[ranks[t.rank_off + i] for i in range(t.n_ranks)]) for t in tags[:n]]


def test_mm_tokeniser_known_answers(L):
    # src/mod_bam.rs:1924-1953 (test_delta_list_to_positions): read ACCGCCGTCGTCG, the C's sit at 1,2,4,5,8,11.
    # delta lists -> k-th occurrence ranks; positions = occurrence[rank], checked here against the reference's expected positions
    cpos = [i for i, c in enumerate("ACCGCCGTCGTCG") if c == "C"]
    for ds, expected in (([1, 1, 0], [2, 5, 8]), ([3, 0, 0], [5, 8, 11]), ([3, 1], [5, 11])):
        tags = mm_ranks(L, "C+m?,%s;" % ",".join(map(str, ds)), 13, len(ds))
        assert len(tags) == 1 and tags[0]["base"] == "C" and tags[0]["mode"] == "?" and tags[0]["codes"] == [ord("m")]
        assert [cpos[r] for r in tags[0]["ranks"]] == expected
    # header forms (MmTagInfo::parse, 909-1000): combined codes, ChEBI, strands, modes, whitespace, several tags
    tags = mm_ranks(L, "C+hm?,0,1,0;A-a.,2 ,0;N+76792,5;T+g;", 40, 6 + 2 + 1)
    assert [t["base"] for t in tags] == ["C", "A", "N", "T"] and [t["neg"] for t in tags] == [False, True, False, False]
    assert [t["mode"] for t in tags] == ["?", ".", " ", " "]
    assert tags[0]["codes"] == [ord("h"), ord("m")] and tags[2]["codes"] == [0x80000000 | 76792] and tags[3]["ranks"] == []
    assert tags[0]["ranks"] == [0, 2, 3] and tags[1]["ranks"] == [2, 3] and tags[2]["ranks"] == [5]
    # the reference's duplex and N-base vectors (test_duplex_modbase_info 2512-2569, test_delta_list_converter_n_base 2776-2785)
    d2 = "GACTCGACTGGACGTCGA"
    tags = mm_ranks(L, "C+h?,1,1,0;C+m?,1,1,0;G-h?,1,2,0;G-m?,1,2,0", len(d2), 12)
    occ = {b: [i for i, c in enumerate(d2) if c == b] for b in "CG"}
    assert [[occ[t["base"]][r] for r in t["ranks"]] for t in tags] == [[4, 12, 15], [4, 12, 15], [5, 13, 16], [5, 13, 16]]
    assert [t["neg"] for t in tags] == [False, False, True, True]
    tags = mm_ranks(L, "N+b?,5,0,0,1,3,0,0;", 17, 7)
    assert tags[0]["ranks"] == [5, 6, 7, 9, 13, 14, 15]          # N tags: absolute forward positions
    tags = mm_ranks(L, "C+h?;C+m?;", 15, 0)
    assert [t["ranks"] for t in tags] == [[], []]                  # test_mod_bam_modbase_info_empty: headers without calls
    # rejected by the reference: bad base, missing strand, ML shorter than the calls (1222-1228), N-tag position past the read end
    assert mm_ranks(L, "X+m?,1;", 10, 1) == -1
    assert mm_ranks(L, "Cm?,1;", 10, 1) == -1
    assert mm_ranks(L, "C+m?,1,2,3;", 10, 2) == -1
    assert mm_ranks(L, "N+m?,4,5;", 10, 2) == -1


def test_fxhashmap_iteration_order(L):
    # src/mod_bam.rs:2250-2258 pins h before m whatever the insertion order (FxHasher: Code(c) -> bucket c & 3 in a 4-bucket table)
    L.mkp_host_map_order.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]

    def order(codes):
        a = (ctypes.c_uint32 * len(codes))(*codes)
        o = (ctypes.c_uint32 * 16)()
        n = L.mkp_host_map_order(a, len(codes), o)
        return [o[i] for i in range(n)]
    h, m, a_, c_ = ord("h"), ord("m"), ord("a"), ord("c")
    assert order([h, m]) == [h, m] and order([m, h]) == [h, m]
    assert order([m]) == [m] and order([c_, h]) == [h, c_]          # buckets: h -> 0, m -> 1, a -> 1, c -> 3
    assert order([m, a_]) == [m, a_] and order([a_, m]) == [a_, m]  # same bucket: linear probing keeps insertion order
    assert len(order([h, m, a_, c_])) == 4                          # fourth insert grows the table to 8 buckets
