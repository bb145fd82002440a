"""GPU: inputs outside the device path's coverage must fail loudly (MKP_E_UNSUPPORTED), never fall back or silently
diverge from the reference: columns deeper than --max-depth (htslib's read dropping is not restated), two kept records
sharing a read name in one shard (the reference keys its per-interval cache by name, read_cache.rs:28-35), and flags
whose handling lives in the reference's Rust writers."""
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


def test_duplicate_read_names_are_refused(tmp_path):
    seq = "ACGTCGACGTACGCGTACGATCGCGTA" * 4
    aux = aux_z("MM", "C+m?,0,1;") + aux_bc("ML", [200, 30])
    recs = [bam_record(0, 10, 0, "same_name", [(len(seq), "M")], seq, aux), bam_record(0, 40, 0, "same_name", [(len(seq), "M")], seq, aux)]
    bam = str(tmp_path / "dup.bam")
    bgzf_write(bam, bytes(bam_header([("ctg", 1000)])) + b"".join(recs))
    with pytest.raises(modkit_amd.MkpError) as e:
        modkit_amd.pileup([bam, str(tmp_path / "o.bed"), "--no-filtering"])
    assert e.value.status == -3 and "read name" in str(e.value)


@pytest.mark.parametrize("flag", [["--bedgraph"]])
def test_writer_side_flags_are_refused(tmp_path, flag):
    with pytest.raises(modkit_amd.MkpError) as e:
        modkit_amd.pileup([BC, str(tmp_path / "o.bed"), "--no-filtering"] + flag)
    assert e.value.status == -3
