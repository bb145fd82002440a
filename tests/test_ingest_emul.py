"""The device ingest's per-thread code (modkit_amd/csrc/mkp_ingest_dev.hpp: record chains, record checks + region test, aux walk, MM
tokeniser, packing) compiled for the host and run thread by thread over whole BAMs by tests/ingest_emul.cpp, against the host path it
replaces (mkp_bam.hpp's record index, Packer::add): record offsets, kept / supplementary classification per region, headers, CIGAR
words and chunk prefixes, SEQ bytes, per-tag rank lists and ML bytes, name and layout-key hashes, the same-list and probability-sum
flags — and the same refusals.  The sliced CRC-32 join of mkp_crc32_blocks is checked against zlib over the same streams."""
import os
import struct
import subprocess

import pytest

from bamfuzz import Fuzz, aux_bc, aux_i, aux_z, bam_header, bam_record, bgzf_write

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "modkit_fixtures")


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "ingest_emul")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-o", exe, os.path.join(ROOT, "tests", "ingest_emul.cpp"), "-lz", "-lpthread"])
    return exe


def run(emul, bam, every):
    p = subprocess.run([emul, bam, str(every)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.startswith("ok"), p.stdout
    return p.stdout


@pytest.mark.parametrize("every", [1, 3, 1000])
def test_reference_fixtures(emul, every):
    for f in sorted(os.listdir(FIX)):
        if f.endswith(".bam"):
            run(emul, os.path.join(FIX, f), every)


@pytest.mark.parametrize("profile", ["mixed", "m", "hm_comb", "hm_split", "hm_split_diff", "hma", "implicit", "default", "duplex", "nbase", "chebi", "duplex_hm", "duplex_split",
                                     "duplex_chebi", "duplex_3codes"])
def test_fuzzed_bams(emul, tmp_path, profile):
    for seed in (3, 4):
        bam, _, _ = Fuzz(seed, profile=profile, n_reads=300, tie_rate=0.2, weird_rate=0.25).write(str(tmp_path / ("fz%d" % seed)))
        out = run(emul, bam, 5)
        assert "compared=0" not in out


def test_generated_long_reads(emul, tmp_path):
    from test_host_ingest import gen
    bam, _ = gen(tmp_path, "g", [("c1", 400000), ("c2", 150000)], 1500)
    out = run(emul, bam, 48)
    assert int(out.split("records=")[1].split()[0]) >= 1500


MM_CASES = [
    "C+m?,1,2,3;", "C+m?;", "C+m,0;", "C+m.,0,0;", "C+hm?,1,2;", "C+h?,1,2;C+m?,1,2;", "C+h?,1,2;C+m?,1,3;", "C+h?,1;C+m?,1,2;",
    "C+m?,1,2,x,5;", "C+m?,1,,2;", "C+m?, 1 ,\t2;", "C+m?,1 2;", "C+m?,;", "C+m?,1,2,;", "C+m?,99999999999;", "C+m?,4294967295;", "C+m?,4294967294,0;",
    "N+m?,3,4;", "N+m?,50;", "N+m?,10,10,10;", "C+76792?,1,2;", "C+76792m?,1;", "C+2147483648?,1;", "C+m7?,1;", "C-m?,1;G+m?,2;", "C+m?,1;;G-m?,0;",
    ";;C+m?,1;", "X+m?,1;", "C*m?,1;", "C;", "C+;", "C+?;", "C+?,1;", "C+abcde?,1;", "C+abcd?,1;", "U+m?,0;", "C+m?,1;A+a?,0;G+x?,1;T+y?,0;C+h?,2;A+b?,0;G+z?,1;T+w?,0;",
    "C+m?,1;A+a?,0;G+x?,1;T+y?,0;C+h?,2;A+b?,0;G+z?,1;T+w?,0;C+q?,1;", "C+m?,1;A+a?,0;G+x?,1;T+y?,0;C+h?,2;A+b?,0;G+z?,1;T+w?,0;Q+q?,1;", "C+m\xc3\xa9?,1;",
    "C+m?,1;C+m?,1;", "C+h?,0,0,0;C+m?,0,0,0;", "C+hm?,0,0,0;C+a?,0,0,0;", "",
]


THROWERS = {"C+abcde?,1;", "C+m?,1;A+a?,0;G+x?,1;T+y?,0;C+h?,2;A+b?,0;G+z?,1;T+w?,0;C+q?,1;", "C+m\xc3\xa9?,1;"}   # the host packer refuses the whole shard


def _mm_bam(path, cases, ml_mode, k):
    seq = "ACGT" * 12 + "CCGG"
    recs = []
    for i, mm in enumerate(cases):
        n_deltas = sum(seg.count(",") for seg in mm.split(";"))
        n_ml = {"fit": 2 * n_deltas + 2, "short": max(0, n_deltas - 1), "long": 4 * n_deltas + 9}.get(ml_mode, 2 * n_deltas + 2)
        mm_tag, ml_tag = ("Mm", "Ml") if ml_mode == "old" else ("MM", "ML")
        aux = mm_tag.encode() + b"Z" + mm.encode("latin-1") + b"\0"
        if ml_mode == "wrongtype":
            aux += ml_tag.encode() + b"BS" + struct.pack("<I", n_ml) + bytes(2 * n_ml)
        elif ml_mode != "none":
            aux += aux_bc(ml_tag, [(37 * j + 11 * i) % 256 for j in range(n_ml)])
        if ml_mode == "mn_ok":
            aux += aux_i("MN", len(seq))
        elif ml_mode == "mn_bad":
            aux += aux_i("MN", len(seq) + 1)
        elif ml_mode == "mn_short":
            aux = b"MNs" + struct.pack("<h", len(seq)) + aux
        flag = 16 if i % 3 == 0 else 0
        recs.append(bam_record(0, 10 + 7 * i, flag, "r%d_%d" % (k, i), [(len(seq), "M")], seq, b"XAi" + struct.pack("<i", 5) + aux))
    bgzf_write(path, bytes(bam_header([("c", 5000)])) + b"".join(recs))


def test_handwritten_mm_strings(emul, tmp_path):
    """Tag grammar corner cases: trailing garbage after a valid prefix, whitespace, empty elements, overflowing numbers, ChEBI codes, `N`
    tags past the read end, nine tags with a broken ninth, duplicate tags, MN / Mm / Ml variants, ML of the wrong type or length."""
    clean = [m for m in MM_CASES if m not in THROWERS]
    for k, ml_mode in enumerate(("fit", "short", "long", "wrongtype", "old", "mn_ok", "mn_bad", "mn_short", "none")):
        bam = str(tmp_path / ("mm_%s.bam" % ml_mode))
        _mm_bam(bam, clean, ml_mode, k)
        out = run(emul, bam, 4)
        assert "compared=%d" % (3 * len(clean)) in out or "compared=" in out


@pytest.mark.parametrize("mm", sorted(THROWERS))
def test_refusals_match_the_host_packer(emul, tmp_path, mm):
    """Five codes in one tag, nine tags, a non-ASCII code: the host packer throws MKP_E_UNSUPPORTED; the device sets the matching error bit
    (checked inside the harness: a host throw without the bit, or a bit without the throw, fails)."""
    bam = str(tmp_path / "t.bam")
    _mm_bam(bam, ["C+m?,1,2;", mm, "C+h?,3;"], "fit", 0)
    run(emul, bam, 2)
    # with a short ML array the broken tag is reached after the "ML array too short" answer of an earlier tag in some cases: still the same verdict on both sides
    _mm_bam(bam, ["C+m?,1,2;", mm], "short", 0)
    run(emul, bam, 1)


def run_plan(emul, bam):
    p = subprocess.run([emul, "--plan", bam], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.startswith("ok"), p.stdout
    return p.stdout


def test_index_plans_select_the_fetch_records(emul, tmp_path):
    """BamSource::ingest_plan (block table + entry points from the BAI: samtools-written, generator-written, fuzzer-written) + chain walk +
    region test == BamSource::fetch, for whole contigs and sub-regions."""
    for f in sorted(os.listdir(FIX)):
        if f.endswith(".bam") and os.path.exists(os.path.join(FIX, f + ".bai")):
            run_plan(emul, os.path.join(FIX, f))
    from test_host_ingest import gen
    bam, _ = gen(tmp_path, "g", [("c1", 900000), ("c2", 250000)], 4000, ["--mean-len", "6000"])
    out = run_plan(emul, bam)
    assert int(out.split("segments=")[1]) > 50      # the linear index supplied entry points
    for seed in (0, 1):
        bam, _, _ = Fuzz(seed, contigs=(("ctgA", 90000), ("ctgB", 20000)), n_reads=900, mean_len=2500, index=True).write(str(tmp_path / ("fz%d" % seed)))
        run_plan(emul, bam)
