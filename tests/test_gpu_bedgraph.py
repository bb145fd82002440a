"""`modkit pileup --bedgraph` (BedGraphWriter, src/writers.rs:264-381): one file per (partition key, strand, mod code[, motif]); a projection
of the bedMethyl rows — chrom, pos, pos + 1, n_mod as f32 / valid as f32 through Rust's `{}`, valid coverage.  The expected files are made
here from the device's own bedMethyl of the same run (which the golden / oracle tests pin), and the reference's own bedGraph test
(tests/test_pileup.rs:547-628: 24 partitioned files, each equal to the unpartitioned control's file of its code and strand) is restated."""
import os
from collections import defaultdict

import numpy as np
import pytest

import modkit_amd
from pileup_cases import FIX

pytestmark = pytest.mark.gpu

BC = os.path.join(FIX, "bc_anchored_10_reads.sorted.bam")
HAP = os.path.join(FIX, "bc_anchored_10_reads.haplotyped.sorted.bam")
REF = os.path.join(FIX, "CGI_ladder_3.6kb_ref.fa")
STRAND = {"+": "positive", "-": "negative", ".": "combined"}


def rust_f32(n_mod, valid):
    x = np.float32(n_mod) / np.float32(valid)
    return np.format_float_positional(x, unique=True, trim="-")


def project(bedmethyl_text, single_motif_label=None, prefix=None):
    files = defaultdict(list)
    for line in bedmethyl_text.splitlines():
        f = line.split("\t")
        name = f[3].split(",", 1)
        label = name[0] + ("_" + name[1].replace(",", "") if len(name) > 1 else ("_" + single_motif_label if single_motif_label else ""))
        fn = (prefix + "_" if prefix else "") + label + "_" + STRAND[f[5]] + ".bedgraph"
        files[fn].append("%s\t%s\t%s\t%s\t%s\n" % (f[0], f[1], f[2], rust_f32(int(f[11]), int(f[9])), f[9]))
    return {k: "".join(v) for k, v in files.items()}


def read_dir(d):
    return {fn: open(os.path.join(d, fn)).read() for fn in sorted(os.listdir(d)) if fn.endswith(".bedgraph")}


@pytest.mark.parametrize("flags,label,prefix", [
    (["--no-filtering"], None, None),
    (["-p", "0.25"], None, "pre"),
    (["--no-filtering", "--cpg", "--ref", REF], "CG0", None),
    (["--no-filtering", "--cpg", "--ref", REF, "--combine-strands"], "CG0", None),
    (["--no-filtering", "--motif", "CG", "0", "--motif", "CGCG", "2", "--ref", REF], None, None),
    (["--no-filtering", "--combine-mods"], None, None),
], ids=["nofilt", "filt_prefix", "cpg", "cpg_combine_strands", "two_motifs", "combine_mods"])
def test_bedgraph_is_the_projection_of_the_bedmethyl(tmp_path, flags, label, prefix):
    bed, out = str(tmp_path / "a.bed"), str(tmp_path / "bg")
    modkit_amd.pileup([BC, bed] + flags)
    modkit_amd.pileup([BC, out, "--bedgraph"] + flags + (["--prefix", prefix] if prefix else []))
    want = project(open(bed).read(), label, prefix)
    got = read_dir(out)
    assert sorted(got) == sorted(want) and len(got) >= 2
    for fn in want:
        assert got[fn] == want[fn], fn


def test_partition_tags_bedgraph_equals_control(tmp_path):
    # tests/test_pileup.rs:547-628
    control, part = str(tmp_path / "control"), str(tmp_path / "part")
    modkit_amd.pileup([BC, control, "--no-filtering", "--bedgraph"])
    modkit_amd.pileup([HAP, part, "--partition-tag", "RG", "--partition-tag", "HP", "--no-filtering", "--bedgraph"])
    ctl = read_dir(control)
    assert sorted(ctl) == ["h_negative.bedgraph", "h_positive.bedgraph", "m_negative.bedgraph", "m_positive.bedgraph"]
    got = read_dir(part)
    assert len(got) == 24
    for fn, text in got.items():
        parts = fn.replace(".bedgraph", "").split("_")
        assert text == ctl["%s_%s.bedgraph" % (parts[2], parts[3])], fn


def test_bedgraph_refusals(tmp_path):
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup([BC, str(tmp_path / "x"), "--bedgraph", "--with-header"])
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup([BC, "-", "--bedgraph"])
