"""`modkit summary` as counts through the C ABI (mkp_summary: two walks of the reference's sampling schedule, calls classified by the
sampling kernels in summary mode, counted by mkp_summary_accumulate) against the CPU oracle's restatement of summarize.rs — which
reproduces the numbers the reference's own tests assert (tests/test_oracle_golden.py::test_summary_*)."""
import struct

import pytest

from conftest import SOAK

import modkit_amd
from bamfuzz import Fuzz
from pileup_cases import BC, BED, fixture
from test_oracle_golden import run_oracle_summary

pytestmark = pytest.mark.gpu

FLAG_SETS = [
    ["-i", "25", "--no-sampling"],
    [],
    ["--only-mapped", "-p", "0.3"],
    ["--no-filtering", "-i", "40"],
    ["--filter-threshold", "C:0.8", "--mod-thresholds", "h:0.9", "--no-sampling"],
    ["--filter-threshold", "0.7", "-n", "6", "-i", "60"],
    ["--ignore", "h", "--no-sampling", "-i", "25"],
    ["--edge-filter", "50", "--no-sampling", "-i", "25"],
    ["--include-bed", BED, "-i", "100", "-p", "0.2"],
    ["--region", "oligo_1512_adapters", "--no-sampling"],
]


def f32(x):
    return struct.unpack("<f", struct.pack("<f", float(x)))[0]


def same(dev, ora):
    assert dev["total"] == ora["total"] and dev["reads_with"] == ora["reads_with"]
    assert {k: f32(v) for k, v in dev["threshold"].items()} == {k: f32(v) for k, v in ora["threshold"].items()}
    assert dev["rows"] == ora["rows"]


@pytest.mark.parametrize("fi", range(len(FLAG_SETS)))
def test_fixture_summary_matches_oracle(oracle_bin, fi):
    ctx = modkit_amd.Context()
    try:
        same(ctx.summary(fixture(BC), FLAG_SETS[fi]), run_oracle_summary(oracle_bin, fixture(BC), FLAG_SETS[fi]))
    finally:
        ctx.close()


def test_summary_with_and_without_index(tmp_path):
    # tests/test_summary.rs:174-186 (test_summary_with_regions): the same BAM read through its index and — a copy without one — whole
    import shutil
    bare = str(tmp_path / "no_index.bam")
    shutil.copy(fixture(BC), bare)
    ctx = modkit_amd.Context()
    try:
        for flags in (["-i", "25"], ["-i", "25", "--no-sampling"]):
            same(ctx.summary(fixture(BC), flags), ctx.summary(bare, flags))
    finally:
        ctx.close()


def test_summary_implicit_calls_on_device(oracle_bin):
    # the reference's test_summary_implicit_calls numbers (tests/test_summary.rs:133-172), on the device
    ctx = modkit_amd.Context()
    try:
        s = ctx.summary(fixture("single_read.bam"), ["--include-bed", fixture("include_bed_summary_test.bed"), "--no-filtering", "-i", "32", "--no-sampling"])
        assert s["total"] == 1 and s["reads_with"] == {"A": 1} and s["rows"][("A", "-")] == (8, 0)
    finally:
        ctx.close()


@pytest.mark.parametrize("profile", ["hm_split", "hma", "duplex_hm", "chebi", "implicit", "mixed"])
def test_fuzzed_summary_matches_oracle(oracle_bin, tmp_path, profile):
    bam, _, _ = Fuzz(1212 + SOAK, profile=profile, n_reads=400, tie_rate=0.15).write(str(tmp_path / "fz"))
    ctx = modkit_amd.Context()
    try:
        for flags in ([], ["--only-mapped", "-n", "150", "-i", "3000", "-p", "0.25"], ["--no-sampling", "--filter-threshold", "0.75"], ["--no-sampling", "--no-filtering"]):
            same(ctx.summary(bam, flags), run_oracle_summary(oracle_bin, bam, flags))
    finally:
        ctx.close()
