"""GPU parity, part 1: libmkpileup (through its C ABI, mkp_pileup_main) reproduces the reference's golden
bedMethyl files byte-for-byte, and agrees with the CPU oracle on the same inputs.  Cases mirror
/root/reference/tests/test_pileup.rs."""
import subprocess

import pytest

import modkit_amd
from pileup_cases import BC, GOLDEN_CASES, REF, fixture, update_tags_ambiguous

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,flags,bam,golden", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_device_reproduces_reference_golden(tmp_path, name, flags, bam, golden):
    out = str(tmp_path / "out.bed")
    modkit_amd.pileup([fixture(bam), out] + flags)
    assert open(out).read() == open(fixture(golden)).read()


def test_pileup_old_tags(tmp_path):
    # tests/test_pileup.rs:161-192: the reference's regression golden on a real PacBio BAM (old-style tags updated to `C+m?`, =/X CIGARs)
    bam = update_tags_ambiguous(fixture("HG002_small.ch20._other.sorted.bam"), str(tmp_path / "updated.bam"))
    out = str(tmp_path / "out.bed")
    modkit_amd.pileup([bam, out, "--no-filtering", "--only-tabs"])
    assert open(out).read() == open(fixture("pileup-old-tags-regressiontest.methyl.bed")).read()


@pytest.mark.parametrize("to_code", ["76792", "c"])
def test_pileup_chebi_code_same_output(oracle_bin, tmp_path, to_code):
    # tests/test_pileup.rs:373-444: 5hmC renamed to a ChEBI number / another letter (`adjust-mods --convert`, restated test-side): the
    # no-filter golden with the code renamed (sorted as the reference's test sorts), and — unsorted — the oracle's row order
    from pileup_cases import convert_mod_code, chebi_case_expected_rows
    bam = convert_mod_code(fixture(BC), str(tmp_path / "conv.bam"), "h", to_code)
    out, ora = str(tmp_path / "out.bed"), str(tmp_path / "oracle.bed")
    flags = ["-i", "25", "--no-filtering", "--only-tabs"]
    modkit_amd.pileup([bam, out] + flags)
    want, key = chebi_case_expected_rows(fixture("modbam.modpileup_nofilt.methyl.bed"), to_code)
    got = ["\t".join(f) for f in sorted((l.split("\t") for l in open(out).read().splitlines()), key=key)]
    assert got == want
    p = subprocess.run([oracle_bin, "pileup", bam, ora] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(out).read() == open(ora).read()


def test_pileup_collapse(oracle_bin, tmp_path):
    # tests/test_pileup.rs:91-141: `pileup --ignore h` == `adjust-mods --ignore h` (restated test-side) + plain pileup, -i 25 —
    # on the device, and both against the oracle
    from pileup_cases import collapse_ignore
    bam = collapse_ignore(fixture(BC), str(tmp_path / "collapsed.bam"), "h")
    a, b, ora = str(tmp_path / "collapsed.bed"), str(tmp_path / "restricted.bed"), str(tmp_path / "oracle.bed")
    modkit_amd.pileup([bam, a, "-i", "25", "--no-filtering"])
    modkit_amd.pileup([fixture(BC), b, "-i", "25", "--ignore", "h", "--no-filtering"])
    assert open(a).read() and open(a).read() == open(b).read()
    p = subprocess.run([oracle_bin, "pileup", fixture(BC), ora, "-i", "25", "--ignore", "h", "--no-filtering"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(b).read() == open(ora).read()


def test_preset_traditional_same_as_options(tmp_path):
    # tests/test_pileup.rs:446-487: `--preset traditional` is `--cpg --ignore h --combine-strands`
    a, b = str(tmp_path / "preset.bed"), str(tmp_path / "options.bed")
    modkit_amd.pileup([fixture(BC), a, "--no-filtering", "--mixed-delim", "--preset", "traditional", "--ref", REF])
    modkit_amd.pileup([fixture(BC), b, "--cpg", "--no-filtering", "--mixed-delim", "--ignore", "h", "--combine-strands", "--ref", REF])
    assert open(a).read() == open(b).read() and open(a).read()


def _both(oracle_bin, tmp_path, bam, flags):
    a, b = str(tmp_path / "dev.bed"), str(tmp_path / "oracle.bed")
    modkit_amd.pileup([fixture(bam), a] + flags)
    p = subprocess.run([oracle_bin, "pileup", fixture(bam), b] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return open(a).read(), open(b).read()


def test_duplicated_reads_ignored(tmp_path):
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    modkit_amd.pileup([fixture("duplicated.marked.fixed.bam"), a, "--no-filtering"])
    modkit_amd.pileup([fixture(BC), b, "--no-filtering"])
    assert open(a).read() == open(b).read() and open(a).read()


def test_no_mod_calls(tmp_path):
    out = str(tmp_path / "o.bed")
    modkit_amd.pileup([fixture("empty-tags.sorted.bam"), out, "--no-filtering"])
    assert open(out).read() == ""


@pytest.mark.parametrize("flags", [
    [],  # BASELINE config C1: defaults (sampled 10th-percentile threshold)
    ["--no-filtering", "--force-allow-implicit"],
    ["--filter-threshold", "0.7"],
    ["--filter-threshold", "C:0.8", "--mod-thresholds", "h:0.9"],
    ["--ignore", "h", "--no-filtering"],
    ["--combine-mods", "--filter-threshold", "0.75"],
    ["--preset", "traditional", "--ref", REF],
    ["--cpg", "--ref", REF, "-i", "13", "--no-filtering"],
    ["--edge-filter", "10,30", "--invert-edge-filter", "--no-filtering"],
], ids=lambda f: " ".join(f) or "defaults")
def test_device_equals_oracle_on_bc_anchored(oracle_bin, tmp_path, flags):
    dev, ora = _both(oracle_bin, tmp_path, BC, flags)
    assert dev == ora and len(dev) > 0


def test_device_equals_oracle_old_tags_implicit(oracle_bin, tmp_path):
    # PacBio Mm/Ml tags without a mode character: every unlisted C is an implicit canonical call
    dev, ora = _both(oracle_bin, tmp_path, "HG002_small.ch20._other.sorted.bam", ["--no-filtering", "--force-allow-implicit"])
    assert dev == ora and len(dev.splitlines()) > 40000


def test_device_equals_oracle_old_tags_rejected(oracle_bin, tmp_path):
    # without --force-allow-implicit those reads are coverage-only (read_cache.rs:122-137)
    dev, ora = _both(oracle_bin, tmp_path, "HG002_small.ch20._other.sorted.bam", ["--no-filtering"])
    assert dev == ora == ""


def test_device_equals_oracle_duplex_defaults(oracle_bin, tmp_path):
    dev, ora = _both(oracle_bin, tmp_path, "duplex_modbam.sorted.bam", ["--region", "chr17"])
    assert dev == ora and len(dev) > 0


def _bed_files(d):
    import os
    return sorted(f for f in os.listdir(d) if f.endswith(".bed"))


@pytest.mark.parametrize("flags", [["--no-filtering"], ["--combine-strands", "--ref", REF, "--cpg", "--no-filtering"]], ids=["partitioned", "combine_strands"])
def test_pileup_partition_tags(tmp_path, flags):
    # tests/test_pileup.rs:501-545 and 692-736: the haplotyped fixture holds every read six times, once per (RG, HP) pair;
    # partitioning on RG and HP must give 6 files, each equal to the unpartitioned pileup of the plain fixture
    control, out_dir = str(tmp_path / "control.bed"), str(tmp_path / "parts")
    modkit_amd.pileup([fixture(BC), control] + flags)
    modkit_amd.pileup([fixture("bc_anchored_10_reads.haplotyped.sorted.bam"), out_dir, "--partition-tag", "RG", "--partition-tag", "HP"] + flags)
    files = _bed_files(out_dir)
    assert files == ["A_1.bed", "A_2.bed", "B_1.bed", "B_2.bed", "C_1.bed", "C_2.bed"]
    want = open(control).read()
    assert want
    for f in files:
        assert open(str(tmp_path / "parts" / f)).read() == want, f


def test_partition_tags_vs_oracle_on_split_bams(oracle_bin, tmp_path):
    # generator reads carry HP:i in {1,2,3} or no HP tag: every partition file must equal the oracle's unpartitioned pileup of
    # the reads of that key alone (split test-side), "ungrouped" holding the untagged reads; --prefix names the files
    import gzip
    import json
    import os
    import struct
    from bamfuzz import bgzf_write
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prefix = str(tmp_path / "p")
    subprocess.check_output([os.path.join(root, "tools", "gen_modbam"), "--out", prefix, "--contig", "chrP:300000", "--reads", "3000", "--seed", "77", "--style", "hm",
                             "--mean-len", "5000", "--partition-tag", "HP:3"])
    d = gzip.open(prefix + ".bam").read()
    o = 4
    lt, = struct.unpack_from("<i", d, o); o += 4 + lt
    nr, = struct.unpack_from("<i", d, o); o += 4
    for _ in range(nr):
        ln, = struct.unpack_from("<i", d, o); o += 4 + ln + 4
    header, groups = d[:o], {}
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        rec = d[o:o + 4 + bs]; o += 4 + bs
        k = rec.rfind(b"HPi")   # the generator appends the tag last
        key = str(struct.unpack_from("<i", rec, k + 3)[0]) if k == len(rec) - 7 else "ungrouped"
        groups.setdefault(key, bytearray(header)).extend(rec)
    assert sorted(groups) == ["1", "2", "3", "ungrouped"]
    flags = ["--filter-threshold", "0.7", "--cpg", "--ref", prefix + ".fa"]
    out_dir = str(tmp_path / "parts")
    modkit_amd.pileup([prefix + ".bam", out_dir, "--partition-tag", "HP", "--prefix", "hap"] + flags)
    assert _bed_files(out_dir) == ["hap_1.bed", "hap_2.bed", "hap_3.bed", "hap_ungrouped.bed"]
    for key, data in groups.items():
        sub = str(tmp_path / ("sub_%s.bam" % key))
        bgzf_write(sub, bytes(data))
        ora = str(tmp_path / ("ora_%s.bed" % key))
        p = subprocess.run([oracle_bin, "pileup", sub, ora] + flags, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        got, want = open(os.path.join(out_dir, "hap_%s.bed" % key)).read(), open(ora).read()
        assert want and got == want, key


def test_bgzf_output_holds_the_golden_text(tmp_path):
    # --bgzf: what `bgzip` + `tabix -p bed` would make of the reference's output — the gzip stream is the golden bedMethyl, the
    # .tbi is there (its contents are checked against region scans in tests/test_format_cpu.py)
    import gzip
    from pileup_cases import HEMI_GOLDEN_CASES, hemi_reference_fasta
    name, flags, bam, golden = GOLDEN_CASES[0]
    out = str(tmp_path / "out.bed.gz")
    modkit_amd.pileup([fixture(bam), out] + flags + ["--bgzf"])
    assert gzip.decompress(open(out, "rb").read()) == open(fixture(golden), "rb").read()
    assert open(out + ".tbi", "rb").read()[:4] == b"\x1f\x8b\x08\x04"
    name, flags, bam, golden = HEMI_GOLDEN_CASES[0]
    out = str(tmp_path / "hemi.bed.gz")
    modkit_amd.pileup_hemi([fixture(bam), "-o", out] + flags + ["-r", hemi_reference_fasta(tmp_path), "--bgzf"])
    assert gzip.decompress(open(out, "rb").read()) == open(fixture(golden), "rb").read()
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup([fixture(bam), str(tmp_path / "x.gz"), "--no-filtering", "--bgzf", "--with-header"])
