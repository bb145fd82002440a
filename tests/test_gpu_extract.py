"""GPU parity, part 7: `extract calls` (mkp_extract_calls_main: calls decoded on the device, the table assembled on the host) against the
reference's own golden tables (tests/test_extract.rs:499-560) byte for byte, and against the oracle on fuzzed modBAMs (every tag
layout of tests/bamfuzz.py, filters, thresholds)."""
import subprocess

import pytest

from conftest import SOAK

import modkit_amd
from bamfuzz import Fuzz
from pileup_cases import EXTRACT_CALLS_CASES, REF, fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,flags,bam,golden", EXTRACT_CALLS_CASES, ids=[c[0] for c in EXTRACT_CALLS_CASES])
def test_device_reproduces_extract_calls_golden(tmp_path, name, flags, bam, golden):
    out = str(tmp_path / "calls.tsv")
    modkit_amd.extract_calls([fixture(bam), out] + [f.format(ref=REF) for f in flags])
    assert open(out).read() == open(fixture(golden)).read()


FLAG_SETS = [
    ["--no-filtering"],
    ["--filter-threshold", "0.75", "--mod-thresholds", "m:0.9"],
    ["--filter-threshold", "C:0.7", "--filter-threshold", "0.6", "--pass-only"],
    ["--ignore", "h", "--filter-threshold", "0.7"],
    ["--edge-filter", "12,30", "--no-filtering"],
    ["--mapped-only", "--no-filtering", "--kmer-size", "8"],
    ["--allow-non-primary", "--no-filtering", "--ref", "{fa}"],
    ["-p", "0.2", "--ref", "{fa}"],
    [],
    # round 6: --include-bed / --region / --num-reads / --ignore-index (src/extract/subcommand.rs:540-560, util.rs:126-160, 329-575)
    ["--include-bed", "{bed}", "--no-filtering"],
    ["--region", "ctgA:2000-7000", "--filter-threshold", "0.7"],            # indexed input: the records overlapping the region
    ["--region", "ctgA:2000-7000", "--ignore-index", "--no-filtering"],     # serial scan: the region only steers the estimate
    ["--num-reads", "57", "--ignore-index", "--no-filtering"],              # the first 57 records that reach process_record
    ["--num-reads", "40", "--no-filtering"],                                # indexed: the sampling schedule, one RecordSampler per interval
    ["--num-reads", "25", "-i", "700", "--region", "ctgA:1000-9000", "--filter-threshold", "0.7"],
    ["--num-reads", "60", "--mapped-only", "-t", "2", "-i", "500", "--no-filtering"],
    ["--num-reads", "1", "--no-filtering", "--allow-non-primary"],
    ["--num-reads", "400", "-i", "2500", "-p", "0.25"],                     # more than the file holds: quotas capped by the index counts
    ["--num-reads", "30", "--include-bed", "{bed}", "--no-filtering"],      # indexed + BED: the schedule over the BED-optimised reference records
    ["--num-reads", "45", "--include-bed", "{bed}", "-i", "300", "--region", "ctgA", "--filter-threshold", "0.6"],
    ["--cpg", "--ref", "{fa}", "--no-filtering"],                           # the include filter = the motif hits of the whole contigs (util.rs:157-277)
    ["--motif", "CG", "0", "--motif", "GATC", "1", "--ref", "{fa}", "--include-bed", "{bed}", "-p", "0.3"],   # ... intersected with the BED; estimate under it
    ["--cpg", "--ref", "{fa}", "--mask", "--num-reads", "40", "--filter-threshold", "0.7"],                     # ... soft-masked bases do not match; the schedule
    ["--exclude-bed", "{bed}", "--no-filtering"],                           # keep = include hit && !exclude hit; rows without a reference position stay
    ["--exclude-bed", "{bed}", "--mapped-only", "--filter-threshold", "0.7", "--num-reads", "80"],
    ["--ignore-implicit", "--no-filtering"],                                # honoured by the reference's interval path only (an index, no --ignore-index)
    ["--ignore-implicit", "--ignore-index", "--no-filtering"],              # ... and silently not by its serial scan
    ["--ignore-implicit", "--num-reads", "50", "--filter-threshold", "0.7"],
    ["--include-bed", "{bed}", "--region", "ctgA", "-p", "0.3", "--mapped-only"],   # estimate under BED + region
]


@pytest.mark.parametrize("profile", ["m", "hm_comb", "hm_split", "hm_split_diff", "hma", "implicit", "default", "nbase", "chebi", "mixed"])
def test_extract_calls_fuzz_vs_oracle(oracle_bin, tmp_path, profile):
    bam, fa, bed = Fuzz(900 + SOAK, profile=profile, n_reads=300).write(str(tmp_path / "fz"), bed=True)
    for fi, fl in enumerate(FLAG_SETS):
        flags = [f.format(fa=fa, bed=bed) for f in fl]
        dev, ora = str(tmp_path / ("dev%d.tsv" % fi)), str(tmp_path / ("ora%d.tsv" % fi))
        p = subprocess.run([oracle_bin, "extract-calls", bam, ora] + flags, capture_output=True, text=True)
        try:
            modkit_amd.extract_calls([bam, dev] + flags)
            err = None
        except modkit_amd.MkpError as e:
            err = e
        if p.returncode != 0:
            assert err is not None, "oracle failed (%s) but the device run succeeded" % p.stderr[-200:]
            continue
        assert err is None, "device failed: %s (flags %s)" % (err, flags)
        a, b = open(dev).read().splitlines(), open(ora).read().splitlines()
        for i in range(max(len(a), len(b))):
            x, y = (a[i] if i < len(a) else "<none>"), (b[i] if i < len(b) else "<none>")
            assert x == y, "profile %s flags %s row %d differs\n device: %s\n oracle: %s (%d vs %d rows)" % (profile, flags, i, x, y, len(a), len(b))
        assert len(a) > 1 or "--include-bed" in flags


def test_bgzf_table_is_the_plain_table_compressed(tmp_path):
    # --bgzf (src/extract/subcommand.rs:629-660): the same rows as BGZF blocks — a multi-member gzip stream closed by the empty EOF block
    import gzip
    bam, fa, _ = Fuzz(907 + SOAK, profile="hm_split", n_reads=300).write(str(tmp_path / "fz"))
    plain, packed = str(tmp_path / "p.tsv"), str(tmp_path / "p.tsv.gz")
    flags = ["--filter-threshold", "0.7", "--ref", fa]
    modkit_amd.extract_calls([bam, plain] + flags)
    modkit_amd.extract_calls([bam, packed, "--bgzf", "--out-threads", "2"] + flags)
    raw = open(packed, "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"
    assert raw[-28:] == bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    text = gzip.open(packed, "rb").read()
    assert text == open(plain, "rb").read() and len(text) > 3 * 0xff00   # several blocks
