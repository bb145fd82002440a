"""GPU parity, part 1: libmkpileup (through its C ABI, mkp_pileup_main) reproduces the reference's golden
bedMethyl files byte-for-byte, and agrees with the CPU oracle on the same inputs.  Cases mirror
/root/reference/tests/test_pileup.rs."""
import subprocess

import pytest

import modkit_amd
from pileup_cases import BC, GOLDEN_CASES, REF, fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,flags,bam,golden", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_device_reproduces_reference_golden(tmp_path, name, flags, bam, golden):
    out = str(tmp_path / "out.bed")
    modkit_amd.pileup([fixture(bam), out] + flags)
    assert open(out).read() == open(fixture(golden)).read()


def _both(oracle_bin, tmp_path, bam, flags):
    a, b = str(tmp_path / "dev.bed"), str(tmp_path / "oracle.bed")
    modkit_amd.pileup([fixture(bam), a] + flags)
    p = subprocess.run([oracle_bin, "pileup", fixture(bam), b] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return open(a).read(), open(b).read()


def test_duplicated_reads_ignored(tmp_path):
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    modkit_amd.pileup([fixture("duplicated.marked.fixed.bam"), a, "--no-filtering"])
    modkit_amd.pileup([fixture(BC), b, "--no-filtering"])
    assert open(a).read() == open(b).read() and open(a).read()


def test_no_mod_calls(tmp_path):
    out = str(tmp_path / "o.bed")
    modkit_amd.pileup([fixture("empty-tags.sorted.bam"), out, "--no-filtering"])
    assert open(out).read() == ""


@pytest.mark.parametrize("flags", [
    [],  # BASELINE config C1: defaults (sampled 10th-percentile threshold)
    ["--no-filtering", "--force-allow-implicit"],
    ["--filter-threshold", "0.7"],
    ["--filter-threshold", "C:0.8", "--mod-thresholds", "h:0.9"],
    ["--ignore", "h", "--no-filtering"],
    ["--combine-mods", "--filter-threshold", "0.75"],
    ["--preset", "traditional", "--ref", REF],
    ["--cpg", "--ref", REF, "-i", "13", "--no-filtering"],
    ["--edge-filter", "10,30", "--invert-edge-filter", "--no-filtering"],
], ids=lambda f: " ".join(f) or "defaults")
def test_device_equals_oracle_on_bc_anchored(oracle_bin, tmp_path, flags):
    dev, ora = _both(oracle_bin, tmp_path, BC, flags)
    assert dev == ora and len(dev) > 0


def test_device_equals_oracle_old_tags_implicit(oracle_bin, tmp_path):
    # PacBio Mm/Ml tags without a mode character: every unlisted C is an implicit canonical call
    dev, ora = _both(oracle_bin, tmp_path, "HG002_small.ch20._other.sorted.bam", ["--no-filtering", "--force-allow-implicit"])
    assert dev == ora and len(dev.splitlines()) > 40000


def test_device_equals_oracle_old_tags_rejected(oracle_bin, tmp_path):
    # without --force-allow-implicit those reads are coverage-only (read_cache.rs:122-137)
    dev, ora = _both(oracle_bin, tmp_path, "HG002_small.ch20._other.sorted.bam", ["--no-filtering"])
    assert dev == ora == ""


def test_device_equals_oracle_duplex_defaults(oracle_bin, tmp_path):
    dev, ora = _both(oracle_bin, tmp_path, "duplex_modbam.sorted.bam", ["--region", "chr17"])
    assert dev == ora and len(dev) > 0
