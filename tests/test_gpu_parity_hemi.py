"""GPU parity for `pileup-hemi` (duplex pattern counts; the reference's src/pileup/duplex.rs): the device path through the
C ABI against (1) the reference's own goldens (tests/test_pileup_hemi.rs), (2) the CPU oracle on the same fixture under other
flags, (3) the oracle on fuzzed duplex modBAMs — deletions, ref-skips, SNPs at the motif, records whose tags fail, many small
intervals and tiles.  Whole-file, byte-exact comparison."""
import subprocess

import pytest

from conftest import SOAK

import modkit_amd
from bamfuzz import Fuzz
from pileup_cases import HEMI_BAM, HEMI_GOLDEN_CASES, fixture, hemi_reference_fasta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hemi_ref(tmp_path_factory):
    return hemi_reference_fasta(tmp_path_factory.mktemp("hemi_ref"))


def first_diff(a, b):
    al, bl = a.splitlines(), b.splitlines()
    for i in range(max(len(al), len(bl))):
        x = al[i] if i < len(al) else "<none>"
        y = bl[i] if i < len(bl) else "<none>"
        if x != y:
            return "row %d differs\n device: %s\n oracle: %s\n (%d vs %d rows)" % (i, x, y, len(al), len(bl))
    return None


@pytest.mark.parametrize("name,flags,bam,golden", HEMI_GOLDEN_CASES, ids=[c[0] for c in HEMI_GOLDEN_CASES])
def test_device_reproduces_reference_hemi_golden(tmp_path, hemi_ref, name, flags, bam, golden):
    out = str(tmp_path / "out.bed")
    modkit_amd.pileup_hemi([fixture(bam), "-o", out] + flags + ["-r", hemi_ref])
    got, want = open(out).read(), open(fixture(golden)).read()
    assert got == want, first_diff(got, want)


def run_both(oracle_bin, tmp_path, bam, flags, extra_dev=()):
    dev, ora = str(tmp_path / "dev.bed"), str(tmp_path / "ora.bed")
    p = subprocess.run([oracle_bin, "pileup-hemi", bam, "-o", ora] + flags, capture_output=True, text=True)
    try:
        modkit_amd.pileup_hemi([bam, "-o", dev] + flags + list(extra_dev))
        dev_err = None
    except modkit_amd.MkpError as e:
        dev_err = e
    if p.returncode != 0:
        assert dev_err is not None, "oracle failed (%s) but the device run succeeded" % p.stderr[-200:]
        return None
    assert dev_err is None, "device failed: %s" % dev_err
    a, b = open(dev).read(), open(ora).read()
    assert a == b, first_diff(a, b)
    return a


_REGION = ["--region", "chr20:22,613,835-22,640,468"]
FIXTURE_FLAGS = [
    ["--cpg", "--no-filtering", "--region", "chr20"],                   # the whole contig: every interval, most of them without reads
    ["--cpg", "--no-filtering", "--combine-mods"] + _REGION,
    ["--cpg", "--ignore", "h", "--filter-threshold", "0.8"] + _REGION,
    ["--cpg", "--filter-threshold", "C:0.9", "--mod-thresholds", "h:0.95", "-i", "1000"] + _REGION,
    ["--cpg", "--no-filtering", "--edge-filter", "5000,2000", "-i", "3000"] + _REGION,
    ["--motif", "CG", "1", "--no-filtering", "-i", "2500"] + _REGION,   # focus on the G: the partner sits one position before
    ["--motif", "GC", "0", "--no-filtering"] + _REGION,
    ["--motif", "CCGG", "1", "-p", "0.2", "-i", "5000"] + _REGION,
    ["--cpg", "-p", "0.25", "-n", "20", "--region", "chr20"],
]


@pytest.mark.parametrize("fi", range(len(FIXTURE_FLAGS)))
def test_fixture_against_oracle(oracle_bin, tmp_path, hemi_ref, fi):
    out = run_both(oracle_bin, tmp_path, fixture(HEMI_BAM), FIXTURE_FLAGS[fi] + ["-r", hemi_ref], extra_dev=["--tile", "256"] if fi % 2 else [])
    if fi in (0, 1):
        assert out and len(out.splitlines()) > 50


FUZZ_FLAGS = [
    ["--cpg", "--no-filtering"],
    ["--cpg", "--no-filtering", "-i", "333"],
    ["--cpg", "--filter-threshold", "0.7", "-i", "777", "--force-allow-implicit"],
    ["--cpg", "--filter-threshold", "C:0.75", "--mod-thresholds", "m:0.6", "--combine-mods", "-i", "1000"],
    ["--motif", "CG", "1", "--no-filtering", "-i", "900"],
    ["--cpg", "--no-filtering", "-i", "100", "--force-allow-implicit"],
    ["--motif", "CCGG", "1", "--filter-threshold", "0.6", "--edge-filter", "15,40", "-i", "512"],
    ["--cpg", "--ignore", "h", "--filter-threshold", "0.66", "--mask"],
    ["--cpg", "--no-filtering", "--region", "ctgA:1500-7300", "-i", "450"],
    ["--cpg", "--include-bed", "{bed}", "--filter-threshold", "0.7", "-i", "900"],
    ["--cpg"],   # sampled threshold
]


@pytest.mark.parametrize("profile", ["duplex", "duplex_hm", "duplex_split", "duplex_chebi", "duplex_3codes", "mixed"])
@pytest.mark.parametrize("fi", range(len(FUZZ_FLAGS)))
def test_fuzz_hemi(oracle_bin, tmp_path, profile, fi):
    bam, fa, bed = Fuzz(4200 + fi + SOAK, profile=profile, n_reads=300, tie_rate=0.1).write(str(tmp_path / "fz"), bed=True)
    flags = [f.format(bed=bed) for f in FUZZ_FLAGS[fi]] + ["-r", fa]
    run_both(oracle_bin, tmp_path, bam, flags, extra_dev=["--tile", "256"] if fi % 2 else [])


def test_hemi_refuses_what_the_reference_refuses(tmp_path, hemi_ref):
    with pytest.raises(modkit_amd.MkpError, match="palindromic"):
        modkit_amd.pileup_hemi([fixture(HEMI_BAM), "-o", str(tmp_path / "o.bed"), "--motif", "CGT", "0", "-r", hemi_ref])
    with pytest.raises(modkit_amd.MkpError, match="--cpg or a --motif"):
        modkit_amd.pileup_hemi([fixture(HEMI_BAM), "-o", str(tmp_path / "o.bed"), "-r", hemi_ref])
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup_hemi([fixture(HEMI_BAM), "-o", str(tmp_path / "o.bed"), "--cpg", "--preset", "traditional", "-r", hemi_ref])


def test_hemi_shard_api_and_mode_switch(tmp_path, hemi_ref):
    """mkp_hemi_shard_run on a context that then runs the same shard as a plain pileup: the resident plan follows the mode."""
    ctx = modkit_amd.Context()
    try:
        out = str(tmp_path / "h.bed")
        rep = ctx.pileup_hemi_run([fixture(HEMI_BAM), "-o", out, "--cpg", "--no-filtering", "--mixed-delim", "-r", hemi_ref, "--region", "chr20:22,613,835-22,640,468"])
        assert rep.n_rows == 341 and open(out).read() == open(fixture("duplex_hemi_nofilt.bed")).read()
        ctx.rerun(2)                                  # hemi kernels again on the resident shard
        with pytest.raises(modkit_amd.MkpError):
            ctx.rerun(1, fetch=True)                  # plain rows cannot be read from a hemi plan
        out2 = str(tmp_path / "p.bed")
        ctx.pileup_run([fixture(HEMI_BAM), out2, "--cpg", "--no-filtering", "--ref", hemi_ref, "--region", "chr20:22,613,835-22,640,468"])
        assert len(open(out2).read().splitlines()) > 300
    finally:
        ctx.close()


def test_hemi_rank_sharded_equals_single(oracle_bin, tmp_path):
    """--gpus-rank R --gpus-world W deals runs of shards to the ranks (the plain pileup's plan): the ranks' outputs, concatenated in
    rank order, are the single-rank output."""
    bam, fa, _ = Fuzz(4311, profile="duplex_hm", n_reads=500, contigs=(("ctgA", 30000), ("ctgB", 9000), ("ctgC", 4000))).write(str(tmp_path / "fz"))
    flags = ["--cpg", "-r", fa, "--filter-threshold", "0.7", "-i", "1000"]
    single = run_both(oracle_bin, tmp_path, bam, flags, extra_dev=["--shard-bp", "3000"])
    parts = []
    for r in range(3):
        out = str(tmp_path / ("rank%d.bed" % r))
        modkit_amd.pileup_hemi([bam, "-o", out] + flags + ["--shard-bp", "3000", "--gpus-rank", str(r), "--gpus-world", "3"])
        parts.append(open(out).read())
    assert all(parts) and "".join(parts) == single
