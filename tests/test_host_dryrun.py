"""Host side without a device: `--plan-only` runs ingest (BGZF/BAM), the interval grid / motif / BED focus builder and the
packer (aux scan, MM tokeniser, layout interning) over every shard and reports kept records and listed calls.  Checks
that the host path accepts all of the reference's fixture BAMs and keeps exactly the records the pileup engine would
(htslib's default mask + supplementary, src/pileup/mod.rs:783-791)."""
import os

import pytest

import modkit_amd

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modkit_fixtures")


def dry(bam, flags, tmp_path):
    out = str(tmp_path / "plan.tsv")
    modkit_amd.build()
    modkit_amd.pileup([os.path.join(FIX, bam), out, "--plan-only"] + flags)
    rows = [l.split("\t") for l in open(out).read().splitlines()]
    return [(r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in rows]


def test_bc_anchored_all_ten_reads_packed(tmp_path):
    rows = dry("bc_anchored_10_reads.sorted.bam", [], tmp_path)
    assert len(rows) == 34 and sum(r[3] for r in rows) == 10          # 34 contigs, 10 primary mapped reads
    assert sum(r[4] for r in rows) > 100                               # MM calls listed by the `C+h?;C+m?` tags
    dup = dry("duplicated.marked.fixed.bam", [], tmp_path)
    assert sum(r[3] for r in dup) == 10                                 # DUP / secondary / supplementary copies are dropped


def test_interval_grid_and_focus_builders_run(tmp_path):
    ref = os.path.join(FIX, "CGI_ladder_3.6kb_ref.fa")
    bed = os.path.join(FIX, "CGI_ladder_3.6kb_ref_include_positions.bed")
    a = dry("bc_anchored_10_reads.sorted.bam", ["--cpg", "--ref", ref, "-i", "37"], tmp_path)
    b = dry("bc_anchored_10_reads.sorted.bam", ["--cpg", "--combine-strands", "--ref", ref, "-i", "37"], tmp_path)
    whole = dry("bc_anchored_10_reads.sorted.bam", [], tmp_path)
    total = sum(r[2] - r[1] for r in whole)
    assert total == 9041 and sum(r[2] - r[1] for r in a) == sum(r[2] - r[1] for r in b) == total  # shards tile every contig exactly once
    c = dry("bc_anchored_10_reads.sorted.bam", ["--include-bed", bed], tmp_path)
    assert 0 < len(c) <= 34


@pytest.mark.parametrize("bam", ["duplex_modbam.sorted.bam", "HG002_small.ch20._other.sorted.bam", "empty-tags.sorted.bam"])
def test_other_fixtures_pack(tmp_path, bam):
    rows = dry(bam, [], tmp_path)
    assert rows and all(r[2] > r[1] for r in rows)
    if bam.startswith("empty"):
        assert sum(r[4] for r in rows) == 0   # reads without usable tags: coverage only
    else:
        assert sum(r[4] for r in rows) > 1000


def test_parallel_pack_equals_sequential_pack(tmp_path):
    """mkp_shard_add_records packs large record batches on all cores (per-thread pieces, then ShardHost::append_all):
    the digest of everything handed to the device must equal the one-thread pack's, byte for byte."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bamfuzz import Fuzz
    modkit_amd.build()
    bam, fa, bed = Fuzz(4242, contigs=(("ctgA", 20000), ("ctgB", 5000)), n_reads=700, mean_len=600, profile="mixed").write(str(tmp_path / "fz"))
    outs = []
    for pack_min in ("0", "1000000"):
        out = str(tmp_path / ("plan_%s.tsv" % pack_min))
        modkit_amd.pileup([bam, out, "--plan-only", "--plan-pack-min", pack_min])
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") >= 2
    assert all(len(l.split("\t")) == 6 for l in outs[0].splitlines())


def test_corrupt_bams_end_in_clean_errors(tmp_path):
    """tools/mutate_bam.py (random bytes, truncation, record core fields, bytes near MM/ML/MN) against the plain build: every
    mutated file is either processed or refused with an error code — no crash.  (The same tool runs against the ASan/UBSan build
    of tools/asan_host.sh outside the suite.)"""
    import subprocess
    import sys
    modkit_amd.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "mutate_bam.py"), os.path.join(FIX, "bc_anchored_10_reads.sorted.bam"), "40", "5",
                        os.path.join(root, "modkit_amd", "csrc", "mkpileup")], capture_output=True, text=True)
    assert p.returncode == 0 and "done: bad 0" in p.stdout, p.stdout[-800:] + p.stderr[-800:]


def test_out_of_range_reference_id_is_refused(tmp_path):
    import struct
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bamfuzz import bam_header, bam_record, bgzf_write
    data = bam_header((("c", 1000),)) + bam_record(0, 10, 0, "r1", [(20, "M")], "ACGT" * 5, b"")
    rec0 = len(bam_header((("c", 1000),)))
    bad = bytearray(data)
    struct.pack_into("<i", bad, rec0 + 4, 7)          # refID 7 of 1
    path = str(tmp_path / "bad.bam")
    bgzf_write(path, bytes(bad))
    with pytest.raises(modkit_amd.MkpError) as e:
        modkit_amd.pileup([path, str(tmp_path / "o.tsv"), "--plan-only"])
    assert e.value.status == -2 and "corrupt BAM record" in str(e.value)
