"""Device ingest on the GPU: `pileup` on indexed BAMs uploads the compressed BGZF blocks, inflates them, cuts and filters
the records, tokenises MM / ML and packs the shard in HBM (mkp_ingest.hip; host side mkp_ingest_host.cpp) — and the threshold estimate
samples from that resident shard when it is the run's only one.  The bedMethyl must equal the host ingest's (`--host-ingest`) and the
oracle's, whatever the flags; corrupt input must fail as loudly as on the host path."""
import os
import re
import subprocess

import pytest

from conftest import SOAK

import modkit_amd
from bamfuzz import Fuzz
from test_host_ingest import gen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "modkit_fixtures")
CLI = os.path.join(ROOT, "modkit_amd", "csrc", "mkpileup")


def cli(args, env=None):
    modkit_amd.build()
    return subprocess.run([CLI] + args, capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=900)


def both(tmp_path, sub, bam, flags, env=None):
    """(device-ingest output, host-ingest output, device run's stderr)"""
    outs = []
    for k, extra in enumerate(([], ["--host-ingest"])):
        out = str(tmp_path / ("o%d.bed" % k))
        p = cli([sub, bam] + (["-o", out] if sub == "pileup-hemi" else [out]) + flags + extra + ["--stats"], env)
        assert p.returncode == 0, p.stderr[-1500:]
        outs.append((open(out, "rb").read(), p.stderr))
    assert "device ingest:" in outs[0][1] and "device ingest:" not in outs[1][1]
    return outs[0][0], outs[1][0], outs[0][1]


FLAGS = [
    [],
    ["--cpg", "--ref", "{fa}"],
    ["--preset", "traditional", "--ref", "{fa}", "-p", "0.2"],
    ["--no-filtering", "--force-allow-implicit"],
    ["--filter-threshold", "0.7", "--ignore", "h", "-i", "777"],
    ["--cpg", "--ref", "{fa}", "--combine-strands", "-n", "150"],
    ["--region", "ctgA:3000-9000", "-p", "0.3"],
    ["--include-bed", "{bed}", "--filter-threshold", "0.7"],
    ["--motif", "CG", "0", "--motif", "GATC", "1", "--ref", "{fa}", "-f", "1.0"],
    ["--edge-filter", "10,30", "-f", "0.6", "--sampling-interval-size", "1500"],
    ["--cpg", "--ref", "{fa}", "--shard-bp", "3000", "-i", "500"],
]


@pytest.mark.parametrize("fi", range(len(FLAGS)))
@pytest.mark.parametrize("profile", ["mixed", "hm_split", "duplex"])
def test_device_ingest_equals_host_ingest_and_oracle(oracle_bin, tmp_path, profile, fi):
    bam, fa, bed = Fuzz(31 + fi + SOAK, profile=profile, n_reads=500, weird_rate=0.15, index=True).write(str(tmp_path / "fz"), bed=True)
    flags = [f.format(fa=fa, bed=bed) for f in FLAGS[fi]]
    dev, host, _ = both(tmp_path, "pileup", bam, flags)
    assert dev == host
    ora = str(tmp_path / "ora.bed")
    oflags = []   # (--shard-bp is the device driver's knob, not a flag of the subcommand)
    k = 0
    while k < len(flags):
        if flags[k] == "--shard-bp":
            k += 2
            continue
        oflags.append(flags[k])
        k += 1
    p = subprocess.run([oracle_bin, "pileup", bam, ora] + oflags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-400:]
    assert dev == open(ora, "rb").read()


def test_single_contig_run_samples_from_the_resident_shard(oracle_bin, tmp_path):
    """One contig, default thresholds: the estimate must take the resident route (the stats line says so) and give the oracle's rows; with
    QC-fail reads in the file (candidates of the sampler that the pileup drops) it must fall back to the host sampler — same rows."""
    bam, fa = gen(tmp_path, "one", [("c1", 300000)], 4000, ["--mean-len", "3000"])
    for flags in ([], ["--cpg", "--ref", fa], ["-n", "500", "-p", "0.25"], ["-f", "0.3"], ["--cpg", "--ref", fa, "--region", "c1:50000-220000"]):
        dev, host, err = both(tmp_path, "pileup", bam, flags)
        assert dev == host and len(dev) > 1000
        assert "first shard attached (resident sampling)" in cli(["pileup", bam, str(tmp_path / "t.bed")] + flags, {"MKP_TRACE_PLAN": "1"}).stderr
        ora = str(tmp_path / "ora.bed")
        assert subprocess.run([oracle_bin, "pileup", bam, ora] + flags, capture_output=True).returncode == 0
        assert dev == open(ora, "rb").read()
    # the fuzzer's mixed profile writes QC-fail records: one contig of it
    bam, fa, _ = Fuzz(8, contigs=(("only", 20000),), n_reads=700, index=True).write(str(tmp_path / "q"))
    dev, host, _ = both(tmp_path, "pileup", bam, ["-n", "200"])
    assert dev == host
    ora = str(tmp_path / "ora2.bed")
    assert subprocess.run([oracle_bin, "pileup", bam, ora, "-n", "200"], capture_output=True).returncode == 0
    assert dev == open(ora, "rb").read()


def test_hemi_on_device_ingest(tmp_path):
    from pileup_cases import HEMI_GOLDEN_CASES, fixture, hemi_reference_fasta
    ref = hemi_reference_fasta(tmp_path)
    for name, flags, bam, golden in HEMI_GOLDEN_CASES:
        dev, host, _ = both(tmp_path, "pileup-hemi", fixture(bam), flags + ["-r", ref])
        assert dev == host and dev == open(fixture(golden), "rb").read(), name


@pytest.mark.parametrize("kind", ["crc", "payload"])
def test_corrupt_input_fails_loudly(tmp_path, kind):
    """A flipped CRC word, a flipped payload byte: the device decoder's status / the CRC kernel must turn into an I/O error.  (Record-level
    corruption — block_size, field lengths, ids — is covered by the error bits in the CPU emulation, tests/test_ingest_emul.py.)"""
    bam, _ = gen(tmp_path, "bad", [("c1", 120000)], 800)
    data = bytearray(open(bam, "rb").read())
    o, k = 0, 0
    while o + 18 <= len(data):
        bsize = int.from_bytes(data[o + 16:o + 18], "little") + 1
        if k == 3:
            if kind == "crc":
                data[o + bsize - 8] ^= 0x21
            else:
                data[o + 18 + (bsize - 26) // 2] ^= 0xff
            break
        o += bsize
        k += 1
    bad = str(tmp_path / "bad2.bam")
    open(bad, "wb").write(bytes(data))
    open(bad + ".bai", "wb").write(open(bam + ".bai", "rb").read())
    p = cli(["pileup", bad, str(tmp_path / "o.bed"), "--no-filtering"])
    assert p.returncode != 0
    assert "corrupt BGZF" in p.stderr, p.stderr


def test_reference_fixtures_through_the_device_ingest(tmp_path):
    """Every indexed fixture BAM of the reference, a few flag sets: device ingest == host ingest, and the --stats line reports the blocks
    and records the device handled."""
    ref = os.path.join(FIX, "CGI_ladder_3.6kb_ref.fa")
    for f, flags in (("bc_anchored_10_reads.sorted.bam", ["--cpg", "--ref", ref, "-p", "0.25"]), ("bc_anchored_10_reads.sorted.bam", ["--no-filtering", "-i", "25"]),
                     ("HG002_small.ch20._other.sorted.bam", ["--no-filtering"]), ("duplicated.marked.fixed.bam", ["--no-filtering"]), ("empty-tags.sorted.bam", ["--no-filtering"]),
                     ("CG_5mC_20230207_1700_6A_PAG66026_3c0abf27_oligo_741_adapters_modcalls_0th_sort_10_reads.bam", ["--filter-threshold", "0.8"])):
        dev, host, err = both(tmp_path, "pileup", os.path.join(FIX, f), flags)
        assert dev == host, f
        m = re.search(r"device ingest: (\d+) BGZF blocks, (\d+) records", err)
        assert m and int(m.group(1)) >= 1


def test_window_that_outgrows_its_buffer_is_inflated_again(oracle_bin, tmp_path):
    """the staged ingest sizes the inflated window before it knows its size (6 x the compressed bytes).  A window that outgrows the
    buffer must not be written past it: blocks behind the capacity get no room on the device (mkp_bgzf_layout), later stages only build their
    tables, and the whole window is inflated again into an exact allocation — whichever stage the overflow is in.  The test shrinks the
    upload pieces / stages and forces the capacity (mkp_internal_ingest_tune) so that a few-MB BAM has several stages and overflows in the
    first, a middle and the last of them; rows must equal the host ingest's and the oracle's."""
    import ctypes
    bam, fa = gen(tmp_path, "grow", [("c1", 600000)], 9000, ["--mean-len", "3000"])
    size = os.path.getsize(bam)
    assert size > 5 * (1 << 20)
    flags = ["--cpg", "--ref", fa, "--filter-threshold", "0.7"]
    host = str(tmp_path / "host.bed")
    modkit_amd.pileup([bam, host] + flags + ["--host-ingest"])
    ora = str(tmp_path / "ora.bed")
    assert subprocess.run([oracle_bin, "pileup", bam, ora] + flags, capture_output=True).returncode == 0
    want = open(host, "rb").read()
    assert want == open(ora, "rb").read() and len(want) > 100000
    L = modkit_amd.lib()
    L.mkp_internal_ingest_tune.argtypes = [ctypes.c_uint64] * 3
    L.mkp_internal_ingest_tune.restype = None
    L.mkp_internal_ingest_reinflated.restype = ctypes.c_uint64
    L.mkp_internal_ingest_staged.restype = ctypes.c_uint64
    try:
        # 64 KiB pieces x 16 slots x 1 round = 1 MiB stages: >= 5 stages.  Capacities: below the first stage's output, in the middle of the
        # window, just short of its end (the inflated window of these BAMs is ~3.5 x the file), and one that holds it all (staged path stands)
        for cap, overflow in ((1 << 20, True), (int(1.7 * size), True), (int(3.0 * size), True), (8 * size, False)):
            before, staged = L.mkp_internal_ingest_reinflated(), L.mkp_internal_ingest_staged()
            L.mkp_internal_ingest_tune(65536, 1, cap)
            out = str(tmp_path / ("dev_%d.bed" % cap))
            modkit_amd.pileup([bam, out] + flags)
            assert open(out, "rb").read() == want, cap
            assert (L.mkp_internal_ingest_reinflated() > before) == overflow, cap
            assert (L.mkp_internal_ingest_staged() > staged) == (not overflow), cap
    finally:
        L.mkp_internal_ingest_tune(0, 0, 0)
