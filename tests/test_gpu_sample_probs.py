"""`modkit sample-probs` percentiles through the C ABI (mkp_sample_probs: the reference's sampling schedule on the host, the sampled
reads decoded by the sampling kernels, exact order statistics from the HBM-resident sample) against the CPU oracle's restatement
(percentile_linear_interp on the sorted sample): f32-identical values and the same number of sampled calls."""
import struct
import subprocess

import pytest

import modkit_amd
from bamfuzz import Fuzz
from pileup_cases import BC, BED, fixture
from sampler_cases import unmapped_tail_bam

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


def test_seeded_fraction_of_the_unmapped_reads(oracle_bin, tmp_path):
    # `-f 0.4 --seed S` (record_sampler.rs:29-38, 80-86): the unmapped records that enter the sample follow StdRng's draws — the same on
    # both sides, different from seed to seed, and refused without a seed (the reference would seed from entropy)
    bam = unmapped_tail_bam(str(tmp_path / "un"))
    qs = [0.1, 0.5, 0.9]
    ctx = modkit_amd.Context()
    try:
        ns = []
        for flags in (["-f", "0.4", "--seed", "7"], ["-f", "0.4", "--seed", "8"], ["-f", "0.05", "--seed", "123456789012"], ["-f", "0.999", "--seed", "0"], ["--no-sampling"]):
            want = oracle_table(oracle_bin, bam, flags, qs)
            got = ctx.sample_probs(bam, qs, flags)
            assert want is not None and set(got) == set(want) == {"C"}
            assert got["C"]["n"] == want["C"]["n"]
            assert [f32bits(v) for v in got["C"]["percentiles"].values()] == [f32bits(v) for v in want["C"]["percentiles"]]
            ns.append(got["C"]["n"])
        assert ns[0] != ns[1] and ns[2] < ns[0] < ns[3] <= ns[4]
        assert oracle_table(oracle_bin, bam, ["-f", "0.4"], qs) is None
        with pytest.raises(modkit_amd.MkpError):
            ctx.sample_probs(bam, qs, ["-f", "0.4"])
    finally:
        ctx.close()


def test_a_bam_without_an_index_is_sampled_serially(oracle_bin, tmp_path):
    # reads_sampler/mod.rs:129-158: no schedule — the file in file order under RecordSampler (first N records that yield values, or one seeded
    # draw per record); a region is an error
    bam = unmapped_tail_bam(str(tmp_path / "ser"), n_mapped=260, n_unmapped=200, index=False, contig_len=40000)
    qs = [0.1, 0.5, 0.9]
    ctx = modkit_amd.Context()
    try:
        ns = []
        for flags in (["-n", "40"], ["-n", "300"], ["-n", "40", "--only-mapped"], [], ["-f", "0.3", "--seed", "5"], ["-f", "0.3", "--seed", "6"], ["--no-sampling"],
                      ["-n", "100", "--ignore", "m"]):
            want = oracle_table(oracle_bin, bam, flags, qs)
            got = ctx.sample_probs(bam, qs, flags)
            assert want is not None and set(got) == set(want) == {"C"}, flags
            assert got["C"]["n"] == want["C"]["n"], flags
            assert [f32bits(v) for v in got["C"]["percentiles"].values()] == [f32bits(v) for v in want["C"]["percentiles"]], flags
            ns.append(got["C"]["n"])
        assert ns[0] < ns[1] < ns[6] and ns[3] == ns[6] and ns[4] != ns[5]
        for flags in (["--region", "ctg"], ["-f", "0.3"]):    # the reference's own errors: a region needs an index; an entropy seed has no answer
            assert oracle_table(oracle_bin, bam, flags, qs) is None
            with pytest.raises(modkit_amd.MkpError):
                ctx.sample_probs(bam, qs, flags)
        with pytest.raises(modkit_amd.MkpError):             # which records consume a draw under a position filter is not reproduced: refused
            ctx.sample_probs(bam, qs, ["-f", "0.3", "--seed", "5", "--only-mapped"])
    finally:
        ctx.close()
