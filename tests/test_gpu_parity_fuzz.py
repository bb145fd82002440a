"""GPU parity, part 2: differential tests oracle vs device on seeded fuzzed modBAMs (tests/bamfuzz.py).
Contigs are longer than one LDS tile so tile seams, halos and multi-tile reads are exercised; `--tile 256`
forces many small tiles.  Bit-exact comparison of the whole bedMethyl text."""
import os
import subprocess

import pytest

import modkit_amd
from bamfuzz import Fuzz

pytestmark = pytest.mark.gpu

FLAG_SETS = [
    ["--no-filtering"],
    ["--no-filtering", "--force-allow-implicit"],
    ["--filter-threshold", "0.8", "--force-allow-implicit"],
    ["--filter-threshold", "C:0.7", "--filter-threshold", "A:0.9", "--filter-threshold", "0.6", "--mod-thresholds", "m:0.85", "--mod-thresholds", "a:0.55"],
    ["--cpg", "--ref", "{fa}", "--filter-threshold", "0.75"],
    ["--cpg", "--ref", "{fa}", "--filter-threshold", "0.75", "-i", "777"],
    ["--preset", "traditional", "--ref", "{fa}", "--filter-threshold", "0.66", "-i", "1000"],
    ["--motif", "CG", "0", "--motif", "CCGG", "1", "--motif", "A", "0", "--ref", "{fa}", "--no-filtering", "-i", "501"],
    ["--motif", "CG", "0", "--motif", "GATC", "1", "--combine-strands", "--ref", "{fa}", "--no-filtering", "-i", "640"],
    ["--ignore", "h", "--filter-threshold", "0.7", "--combine-mods"],
    ["--ignore", "m", "--filter-threshold", "0.7", "--force-allow-implicit"],
    ["--combine-mods", "--no-filtering", "--edge-filter", "15,40"],
    ["--include-bed", "{bed}", "--filter-threshold", "0.7", "-i", "900"],
    ["--cpg", "--ref", "{fa}", "--mask", "--no-filtering"],
    ["--region", "ctgA:1500-7300", "--filter-threshold", "0.7"],
    [],  # default sampled threshold
    ["-p", "0.3", "-n", "60"],
    ["-f", "1.0", "-p", "0.2", "--ignore", "h"],
    ["-f", "0.5", "--edge-filter", "20", "--sampling-interval-size", "2000"],
    ["--include-bed", "{bed}", "-p", "0.15", "--sampling-interval-size", "1500", "-t", "2"],
]


def run_both(oracle_bin, tmp_path, bam, flags, extra_dev=()):
    dev, ora = str(tmp_path / "dev.bed"), str(tmp_path / "ora.bed")
    p = subprocess.run([oracle_bin, "pileup", bam, ora] + flags, capture_output=True, text=True)
    try:
        modkit_amd.pileup([bam, dev] + flags + list(extra_dev))
        dev_err = None
    except modkit_amd.MkpError as e:
        dev_err = e
    if p.returncode != 0:
        assert dev_err is not None, "oracle failed (%s) but the device run succeeded" % p.stderr[-200:]
        return None
    assert dev_err is None, "device failed: %s" % dev_err
    a, b = open(dev).read(), open(ora).read()
    if a != b:
        al, bl = a.splitlines(), b.splitlines()
        for i in range(max(len(al), len(bl))):
            x = al[i] if i < len(al) else "<none>"
            y = bl[i] if i < len(bl) else "<none>"
            if x != y:
                raise AssertionError("row %d differs\n device: %s\n oracle: %s\n (%d vs %d rows)" % (i, x, y, len(al), len(bl)))
    return a


# MKP_FUZZ_SEEDS=a:b widens the sweep for one-off soak runs (default: the 6 seeds the suite pins)
_S = os.environ.get("MKP_FUZZ_SEEDS", "0:6").split(":")


@pytest.mark.parametrize("seed", range(int(_S[0]), int(_S[1])))
@pytest.mark.parametrize("fi", range(len(FLAG_SETS)))
def test_fuzz_mixed(oracle_bin, tmp_path, seed, fi):
    bam, fa, bed = Fuzz(1000 + seed).write(str(tmp_path / "fz"), bed=True)
    flags = [f.format(fa=fa, bed=bed) for f in FLAG_SETS[fi]]
    out = run_both(oracle_bin, tmp_path, bam, flags, extra_dev=["--tile", "256"] if seed % 2 else [])
    assert out is None or isinstance(out, str)


_PS = os.environ.get("MKP_FUZZ_PROFILE_SEEDS", "77:78").split(":")


@pytest.mark.parametrize("pseed", range(int(_PS[0]), int(_PS[1])))
@pytest.mark.parametrize("profile", ["m", "hm_comb", "hm_split", "hm_split_diff", "hma", "implicit", "default", "duplex", "nbase", "chebi", "duplex_hm", "duplex_split", "duplex_chebi", "duplex_3codes"])
def test_fuzz_profiles(oracle_bin, tmp_path, profile, pseed):
    bam, fa, bed = Fuzz(pseed, profile=profile, n_reads=400, tie_rate=0.2).write(str(tmp_path / "fz"), bed=True)
    rows = 0
    for flags in (["--no-filtering", "--force-allow-implicit"], ["--filter-threshold", "0.7", "--force-allow-implicit", "--cpg", "--ref", fa],
                  ["--preset", "traditional", "--ref", fa, "-p", "0.2", "--force-allow-implicit"]):
        out = run_both(oracle_bin, tmp_path, bam, flags)
        rows += len(out.splitlines()) if out else 0
    assert rows > 0


def test_deep_long_reads_many_tiles(oracle_bin, tmp_path):
    # one 60 kb contig, long reads: every read spans many tiles
    bam, fa, bed = Fuzz(5, contigs=(("long", 60000),), n_reads=500, mean_len=9000, profile="hm_split", weird_rate=0.0).write(str(tmp_path / "fz"))
    out = run_both(oracle_bin, tmp_path, bam, ["--preset", "traditional", "--ref", fa])
    assert out and len(out.splitlines()) > 500
    out = run_both(oracle_bin, tmp_path, bam, ["--filter-threshold", "0.75"], extra_dev=["--tile", "1024"])
    assert out and len(out.splitlines()) > 30000


@pytest.mark.parametrize("shard_bp", [2000, 6000, 14000])
def test_driver_shards_cut_inside_a_contig(oracle_bin, tmp_path, shard_bp):
    # the driver cuts contigs longer than 2^27 bp into shards at interval boundaries; --shard-bp forces that on a 60 kb contig:
    # rows (and the strand-combined rows whose partner sits across the cut) must not depend on where the shards end
    bam, fa, bed = Fuzz(11, contigs=(("long", 60000), ("short", 3000)), n_reads=600, mean_len=5000, profile="hm_split", weird_rate=0.05).write(str(tmp_path / "fz"), bed=True)
    for flags in (["-i", "2000", "--preset", "traditional", "--ref", fa, "--filter-threshold", "0.7"], ["-i", "2000", "--filter-threshold", "0.8", "--include-bed", bed],
                  ["-i", "1000", "--cpg", "--ref", fa, "--combine-strands", "--no-filtering"]):
        out = run_both(oracle_bin, tmp_path, bam, flags, extra_dev=["--shard-bp", str(shard_bp)])
        assert out and len(out.splitlines()) > 100


@pytest.mark.parametrize("profile", ["m", "hm_split", "hma", "duplex_split"])
def test_ultra_long_reads(oracle_bin, tmp_path, profile):
    # reads of ~10^5 bases: hundreds of decode steps per read, CIGARs of tens of thousands of ops (many 64-op chunks), every
    # tile of the contig crossed by every read
    bam, fa, bed = Fuzz(21, contigs=(("ul", 250000),), n_reads=14, mean_len=120000, profile=profile, weird_rate=0.0).write(str(tmp_path / "fz"))
    out = run_both(oracle_bin, tmp_path, bam, ["--filter-threshold", "0.75"])
    assert out and len(out.splitlines()) > 10000
    out = run_both(oracle_bin, tmp_path, bam, ["--cpg", "--ref", fa, "--combine-strands", "-p", "0.2", "--force-allow-implicit"])
    assert out and len(out.splitlines()) > 1000
