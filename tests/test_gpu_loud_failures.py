"""GPU: inputs outside the device path's coverage must fail loudly (MKP_E_UNSUPPORTED), never fall back or silently
diverge from the reference: piles in which htslib would drop records under --max-depth (the dropping is restated in the oracle only; tests/test_gpu_max_depth.py) and flags whose handling lives in
the reference's Rust writers.  (Two kept records sharing a read name in one interval — refused through round 5 — are reproduced now:
tests/test_gpu_dup_names.py.)"""
import os

import pytest

import modkit_amd
from bamfuzz import aux_bc, aux_z, bam_header, bam_record, bgzf_write

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modkit_fixtures")
BC = os.path.join(FIX, "bc_anchored_10_reads.sorted.bam")


def test_max_depth_exceeded_is_an_error(tmp_path):
    out = str(tmp_path / "o.bed")
    with pytest.raises(modkit_amd.MkpError) as e:
        modkit_amd.pileup([BC, out, "--no-filtering", "--max-depth", "1"])
    assert e.value.status == -3 and "max_depth" in str(e.value)
    modkit_amd.pileup([BC, out, "--no-filtering", "--max-depth", "10"])   # the fixture's deepest column holds fewer reads
    assert open(out).read()


def test_same_name_in_different_intervals_is_two_reads(oracle_bin, tmp_path):
    """The reference's read cache is per interval and keyed by name: mates / split alignments with one name that lie in different intervals
    never meet — the device takes them as two reads (== oracle, which keys its cache the same way); the same two records under an interval
    size that puts them into one interval are still refused."""
    import random
    import subprocess
    rng = random.Random(5)
    ref_len = 6000
    recs = []
    for i in range(60):
        seq = "".join(rng.choice("ACGT") for _ in range(200))
        cs = [k for k, ch in enumerate(seq) if ch == "C"]
        picks = cs[::3][:20]
        deltas, prev = [], -1
        for k, pidx in enumerate(cs):
            if pidx in picks:
                deltas.append(sum(1 for x in cs if prev < x < pidx)); prev = pidx
        aux = aux_z("MM", "C+m?," + ",".join(map(str, deltas)) + ";") + aux_bc("ML", [rng.randrange(256) for _ in deltas])
        pos = rng.randrange(0, 1500) if i % 2 == 0 else rng.randrange(3200, ref_len - 250)
        recs.append((pos, bam_record(0, pos, 0, "pair%02d" % (i // 2), [(len(seq), "M")], seq, aux)))   # mates: one name, far apart
    recs.sort(key=lambda t: t[0])
    bam = str(tmp_path / "mates.bam")
    bgzf_write(bam, bytes(bam_header([("ctg", ref_len)])) + b"".join(r for _, r in recs))
    dev, ora = str(tmp_path / "dev.bed"), str(tmp_path / "ora.bed")
    flags = ["--no-filtering", "-i", "1600"]      # mates sit in intervals 0 and 2-3
    modkit_amd.pileup([bam, dev] + flags)
    assert subprocess.run([oracle_bin, "pileup", bam, ora] + flags, capture_output=True).returncode == 0
    assert open(dev).read() == open(ora).read() and len(open(dev).read()) > 1000
    # one interval for the whole contig: the mates share a cache entry — the later mate is answered from the earlier one's calls (round 6;
    # refused before: tests/test_gpu_dup_names.py)
    flags = ["--no-filtering", "-i", "100000"]
    modkit_amd.pileup([bam, dev] + flags)
    assert subprocess.run([oracle_bin, "pileup", bam, ora] + flags, capture_output=True).returncode == 0
    assert open(dev).read() == open(ora).read() and len(open(dev).read()) > 1000


def test_writer_side_flag_combinations_are_refused(tmp_path):
    # --bedgraph writes a directory of files: no header line, no stdout (subcommand.rs:328-363); the files themselves: tests/test_gpu_bedgraph.py
    for flags in (["--bedgraph", "--with-header"], ["--bedgraph", "--mixed-delim"], ["--bedgraph", "--bgzf"]):
        with pytest.raises(modkit_amd.MkpError) as e:
            modkit_amd.pileup([BC, str(tmp_path / "o"), "--no-filtering"] + flags)
        assert e.value.status == -1, flags
