"""`modkit sample-probs` percentiles through the C ABI (mkp_sample_probs: the reference's sampling schedule on the host, the sampled
reads decoded by the sampling kernels, exact order statistics from the HBM-resident sample) against the CPU oracle's restatement
(percentile_linear_interp on the sorted sample): f32-identical values and the same number of sampled calls."""
import struct
import subprocess

import pytest

import modkit_amd
from bamfuzz import Fuzz
from pileup_cases import BC, BED, fixture

pytestmark = pytest.mark.gpu

FLAG_SETS = [
    [],
    ["--only-mapped"],
    ["--no-sampling", "-i", "25"],
    ["-n", "6", "-i", "40"],
    ["--ignore", "h", "--only-mapped"],
    ["--edge-filter", "30,10", "--no-sampling"],
    ["--include-bed", BED, "-i", "100"],
    ["--region", "oligo_1512_adapters", "--no-sampling"],
]


def oracle_table(oracle_bin, bam, flags, qs):
    p = subprocess.run([oracle_bin, "sample-probs", bam, "-p", ",".join(repr(float(q)) for q in qs)] + flags, capture_output=True, text=True)
    if p.returncode != 0:
        return None
    out = {}
    for ln in p.stdout.splitlines():
        b, q, v, n = ln.split("\t")
        out.setdefault(b, {"n": int(n), "percentiles": []})["percentiles"].append(float(v))
    return out


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


@pytest.mark.parametrize("fi", range(len(FLAG_SETS)))
def test_fixture_percentiles_match_oracle(oracle_bin, fi):
    qs = [0.1, 0.5, 0.9, 0.25, 0.0, 1.0]
    ctx = modkit_amd.Context()
    try:
        want = oracle_table(oracle_bin, fixture(BC), FLAG_SETS[fi], qs)
        try:
            got = ctx.sample_probs(fixture(BC), qs, FLAG_SETS[fi])
        except modkit_amd.MkpError:
            assert want is None
            return
        assert want is not None and set(got) == set(want)
        for b in got:
            assert got[b]["n"] == want[b]["n"]
            assert [f32bits(v) for v in got[b]["percentiles"].values()] == [f32bits(v) for v in want[b]["percentiles"]]
    finally:
        ctx.close()


@pytest.mark.parametrize("profile", ["hm_split", "hma", "duplex_hm", "mixed"])
def test_fuzzed_percentiles_match_oracle(oracle_bin, tmp_path, profile):
    bam, _, _ = Fuzz(909, profile=profile, n_reads=400).write(str(tmp_path / "fz"))
    qs = [0.1, 0.5, 0.9]
    ctx = modkit_amd.Context()
    try:
        for flags in ([], ["--only-mapped", "-n", "150", "-i", "3000"], ["--no-sampling"]):
            want = oracle_table(oracle_bin, bam, flags, qs)
            got = ctx.sample_probs(bam, qs, flags)
            assert want is not None and set(got) == set(want)
            for b in got:
                assert got[b]["n"] == want[b]["n"] and [f32bits(v) for v in got[b]["percentiles"].values()] == [f32bits(v) for v in want[b]["percentiles"]]
    finally:
        ctx.close()
