"""Host ingest without a device (`--plan-only`): the BAI-indexed, bounded reader (mkp_bam.hpp: BaiIndex / BamSource) against the
whole-file loader — same shard plan and the same packed bytes (digest) on the reference's fixtures (indexes written by samtools)
and on generated BAMs (index written by tools/gen_modbam); peak memory that follows the shard, not the file; and a 2-rank plan in
which each rank reads about half of the file."""
import json
import os
import re
import subprocess

import pytest

import modkit_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "modkit_fixtures")
CLI = os.path.join(ROOT, "modkit_amd", "csrc", "mkpileup")


def gen(tmp, name, contigs, reads, extra=()):
    tool = os.path.join(ROOT, "tools", "gen_modbam")
    if not os.path.exists(tool):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)
    prefix = os.path.join(str(tmp), name)
    args = [tool, "--out", prefix, "--reads", str(reads), "--seed", "5", "--style", "hm"] + list(extra)
    for c in contigs:
        args += ["--contig", "%s:%d" % c]
    json.loads(subprocess.check_output(args).decode())
    return prefix + ".bam", prefix + ".fa"


def plan(bam, flags, out):
    """runs the CLI; returns (plan text, stats dict from the --stats line, peak RSS of that process in KiB)"""
    modkit_amd.build()
    p = subprocess.run([CLI, "pileup", bam, out, "--plan-only", "--stats"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    st = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)(?=\s|$)", p.stderr)}
    return open(out).read(), st, st["peak_rss_kb"]


@pytest.mark.parametrize("bam,flags", [
    ("bc_anchored_10_reads.sorted.bam", ["-i", "25"]),
    ("bc_anchored_10_reads.sorted.bam", ["--cpg", "--ref", os.path.join(FIX, "CGI_ladder_3.6kb_ref.fa"), "--region", "oligo_1512_adapters:10-120"]),
    ("duplex_modbam.sorted.bam", ["--region", "chr17"]),
    ("HG002_small.ch20._other.sorted.bam", []),
    ("bc_anchored_10_reads.haplotyped.sorted.bam", []),
])
def test_indexed_fetch_equals_whole_file_on_reference_fixtures(tmp_path, bam, flags):
    a, sa, _ = plan(os.path.join(FIX, bam), flags, str(tmp_path / "a.tsv"))
    b, sb, _ = plan(os.path.join(FIX, bam), flags + ["--no-index"], str(tmp_path / "b.tsv"))
    assert a == b and a
    assert sa["indexed"] == 1 and sb["indexed"] == 0


def test_indexed_fetch_equals_whole_file_on_generated_bam(tmp_path):
    bam, fa = gen(tmp_path, "g", [("chrA", 700_000), ("chrB", 300_000), ("chrC", 120_000)], 6_000, ["--mean-len", "5000"])
    for flags in ([], ["--shard-bp", "100000"], ["--cpg", "--ref", fa, "--shard-bp", "250000"], ["--region", "chrB:100000-250000"]):
        a, sa, _ = plan(bam, flags, str(tmp_path / "a.tsv"))
        b, sb, _ = plan(bam, flags + ["--no-index"], str(tmp_path / "b.tsv"))
        assert a == b and a, flags
        assert sa["indexed"] == 1 and sa["bam_bytes_read"] > 0 and sb["bam_bytes_read"] == 0


def test_peak_memory_follows_the_shard_not_the_file(tmp_path):
    # the same contig shape three times and twelve times over (one shard per contig; two shards are in flight at any time: the next
    # one is fetched while the current one is packed): with the index the peak RSS stays put, without it it grows with the file
    few, _ = gen(tmp_path, "few", [("c%d" % i, 400_000) for i in range(3)], 24_000, ["--mean-len", "5000"])
    many, _ = gen(tmp_path, "many", [("c%d" % i, 400_000) for i in range(12)], 96_000, ["--mean-len", "5000"])
    assert os.path.getsize(many) > 3.5 * os.path.getsize(few)
    t3, s3, rss_few = plan(few, [], str(tmp_path / "p3.tsv"))
    t12, s12, rss_many = plan(many, [], str(tmp_path / "p12.tsv"))
    _, s12w, rss_many_whole = plan(many, ["--no-index"], str(tmp_path / "p12w.tsv"))
    assert len(t3.splitlines()) == 3 and len(t12.splitlines()) == 12 and s12["bam_bytes_inflated"] > 3.5 * s3["bam_bytes_inflated"]
    assert rss_many < 1.3 * rss_few + 32768, (rss_few, rss_many)
    assert rss_many_whole > 2.0 * rss_many, (rss_many, rss_many_whole)      # the whole-file loader holds everything


def test_two_ranks_each_read_about_half_of_the_file(tmp_path):
    bam, _ = gen(tmp_path, "r2", [("c%d" % i, 300_000) for i in range(6)], 18_000, ["--mean-len", "4000"])
    whole, s, _ = plan(bam, ["--shard-bp", "100000"], str(tmp_path / "w.tsv"))
    parts, reads = [], []
    for r in range(2):
        t, st, _ = plan(bam, ["--shard-bp", "100000", "--gpus-rank", str(r), "--gpus-world", "2"], str(tmp_path / ("r%d.tsv" % r)))
        parts.append(t)
        reads.append(st["bam_bytes_read"])
    assert "".join(parts) == whole                                           # contiguous runs, in order, nothing lost
    size = os.path.getsize(bam)
    assert all(0.3 * size < x < 0.7 * size for x in reads), (size, reads)   # balanced by the bytes the index puts under each run
    assert sum(reads) < 1.25 * s["bam_bytes_read"]


def _bgzf_block(data):
    import struct
    import zlib
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def test_oversized_block_size_in_an_indexed_bam_fails_loudly(tmp_path):
    """A record whose block_size runs past the end of the data, reached through the BAI (the bounded reader of mkp_bam.hpp): the old
    reader dropped it silently in the last window (and re-inflated the same window forever in an earlier one); it is an I/O error."""
    import struct
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:100000\n"
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\0" + struct.pack("<i", 100000)

    def rec(pos, block_size=None):
        name = b"r%d\0" % pos
        core = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 30, 4681, 1, 0, 10, -1, -1, 0)
        body = core + name + struct.pack("<I", (10 << 4) | 0) + bytes([0x12] * 5) + bytes([30] * 10)
        return struct.pack("<i", len(body) if block_size is None else block_size) + body
    recs = rec(10) + rec(20, 0x7fffff00) + rec(30)
    b1, b2 = _bgzf_block(hdr), _bgzf_block(recs)
    eof = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    bam = str(tmp_path / "bad.bam")
    open(bam, "wb").write(b1 + b2 + eof)
    beg, end = len(b1) << 16, (len(b1) + len(b2)) << 16
    bai = (b"BAI\1" + struct.pack("<i", 1) + struct.pack("<i", 2) + struct.pack("<Ii", 4681, 1) + struct.pack("<QQ", beg, end) +
           struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", beg, end, 3, 0) +      # htslib's metadata pseudo-bin: without it the reader loads the whole file
           struct.pack("<i", 1) + struct.pack("<Q", beg) + struct.pack("<Q", 0))
    open(bam + ".bai", "wb").write(bai)
    modkit_amd.build()
    p = subprocess.run([CLI, "pileup", bam, str(tmp_path / "o.tsv"), "--plan-only", "--stats"], capture_output=True, text=True, timeout=60)
    assert p.returncode != 0, p.stderr
    assert "BAM record" in p.stderr or "corrupt" in p.stderr, p.stderr


@pytest.mark.parametrize("indexed", [True, False])
def test_crc_mismatch_in_a_block_fails_loudly(tmp_path, indexed):
    """A BGZF block whose payload still inflates to ISIZE bytes but whose CRC32 word does not match (htslib's reader rejects it): an I/O
    error on the indexed reader and on the whole-file loader, never rows."""
    bam, _ = gen(tmp_path, "crc", [("c1", 60000)], 300)
    data = bytearray(open(bam, "rb").read())
    o, k = 0, 0
    while o + 18 <= len(data):   # the CRC word of the third block
        bsize = int.from_bytes(data[o + 16:o + 18], "little") + 1
        if k == 2:
            data[o + bsize - 8] ^= 0x40
            break
        o += bsize
        k += 1
    bad = str(tmp_path / "crc_bad.bam")
    open(bad, "wb").write(bytes(data))
    if indexed:
        open(bad + ".bai", "wb").write(open(bam + ".bai", "rb").read())
    modkit_amd.build()
    p = subprocess.run([CLI, "pileup", bad, str(tmp_path / "o.tsv"), "--plan-only"] + ([] if indexed else ["--no-index"]), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "corrupt BGZF" in p.stderr, p.stderr


@pytest.mark.parametrize("seed", [0, 1, 3])
def test_indexed_fetch_equals_whole_file_on_fuzzed_bams(tmp_path, seed):
    """tests/bamfuzz.py writes its own BAI (bins, linear index, metadata pseudo-bin): reading through it gives the packed bytes of the
    whole-file loader, whole contigs and regions."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bamfuzz import Fuzz
    bam, fa, _ = Fuzz(seed, contigs=(("ctgA", 60000), ("ctgB", 20000)), n_reads=600, mean_len=2500, index=True).write(str(tmp_path / "fz"))
    for flags in ([], ["--shard-bp", "7000"], ["--region", "ctgA:20000-41000"], ["--cpg", "--ref", fa, "-i", "3000", "--shard-bp", "9000"]):
        a, sa, _ = plan(bam, flags, str(tmp_path / "a.tsv"))
        b, sb, _ = plan(bam, flags + ["--no-index"], str(tmp_path / "b.tsv"))
        assert a == b and a, flags
        assert sa["indexed"] == 1 and sb["indexed"] == 0


def test_bed_spans_of_a_contig_share_a_shard(tmp_path):
    """--include-bed: the reference turns every run of BED spans into its own reference record; here the records of one contig share a
    shard (multi-window fetch).  The plan must hold one shard per contig, name the same records through the index and through the
    whole-file loader, and read only a fraction of the file."""
    import random
    bam, fa = gen(tmp_path, "bedm", [("chrA", 3_000_000), ("chrB", 2_000_000)], 12_000, ["--mean-len", "3000"])
    rng = random.Random(3)
    bed = str(tmp_path / "spans.bed")
    with open(bed, "w") as f:
        for name, ln in (("chrA", 3_000_000), ("chrB", 2_000_000)):
            for s in sorted(rng.randrange(0, ln - 3000) for _ in range(10)):
                f.write("%s\t%d\t%d\n" % (name, s, s + 2000))
    a, sa, _ = plan(bam, ["--include-bed", bed], str(tmp_path / "a.tsv"))
    b, sb, _ = plan(bam, ["--include-bed", bed, "--no-index"], str(tmp_path / "b.tsv"))
    assert a == b
    lines = a.strip().split("\n")
    assert len(lines) == 2, lines                       # one merged shard per contig, not one per BED record
    assert sa["bam_bytes_inflated"] < 0.6 * sb["bam_bytes_inflated"]   # the gaps between the spans were not read
    # the same records as one shard per BED record would pack (MKP_NO_BED_MERGE): total kept reads can only differ by reads spanning two records
    env = dict(os.environ, MKP_NO_BED_MERGE="1")
    p = subprocess.run([CLI, "pileup", bam, str(tmp_path / "c.tsv"), "--plan-only", "--include-bed", bed], capture_output=True, text=True, env=env)
    assert p.returncode == 0
    unmerged = open(str(tmp_path / "c.tsv")).read().strip().split("\n")
    assert len(unmerged) > 5
    n_merged = sum(int(l.split("\t")[3]) for l in lines)
    n_unmerged = sum(int(l.split("\t")[3]) for l in unmerged)
    assert n_merged <= n_unmerged and n_merged > 0.5 * n_unmerged
