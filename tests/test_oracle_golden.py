"""Pins the CPU oracle (oracle/, test infrastructure) against the reference's own golden bedMethyl
files: every case of /root/reference/tests/test_pileup.rs whose inputs ship with the reference and that
does not need another subcommand.  Whole-file byte comparison, as check_against_expected_text_file does
(/root/reference/tests/common/mod.rs:113-135)."""
import subprocess

import pytest

from pileup_cases import BC, GOLDEN_CASES, HEMI_BAM, HEMI_GOLDEN_CASES, REF, fixture, hemi_reference_fasta, update_tags_ambiguous


def run_oracle(oracle_bin, bam, out, flags):
    p = subprocess.run([oracle_bin, "pileup", bam, out] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return p.stderr


@pytest.mark.parametrize("name,flags,bam,golden", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_oracle_reproduces_reference_golden(oracle_bin, tmp_path, name, flags, bam, golden):
    out = str(tmp_path / "out.bed")
    run_oracle(oracle_bin, fixture(bam), out, flags)
    assert open(out).read() == open(fixture(golden)).read()


def test_duplicated_reads_ignored(oracle_bin, tmp_path):
    # tests/test_pileup.rs:326 — DUP / secondary / supplementary records do not contribute
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    run_oracle(oracle_bin, fixture("duplicated.marked.fixed.bam"), a, ["--no-filtering"])
    run_oracle(oracle_bin, fixture(BC), b, ["--no-filtering"])
    assert open(a).read() == open(b).read() and len(open(a).read()) > 0


def test_no_mod_calls(oracle_bin, tmp_path):
    # tests/test_pileup.rs:143 — reads without usable tags are coverage-only: no rows
    out = str(tmp_path / "o.bed")
    run_oracle(oracle_bin, fixture("empty-tags.sorted.bam"), out, ["--no-filtering"])
    assert open(out).read() == ""


def test_preset_traditional_same_as_options(oracle_bin, tmp_path):
    # tests/test_pileup.rs:286
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    run_oracle(oracle_bin, fixture(BC), a, ["--preset", "traditional", "--ref", REF, "--no-filtering"])
    run_oracle(oracle_bin, fixture(BC), b, ["--cpg", "--ignore", "h", "--combine-strands", "--ref", REF, "--no-filtering"])
    assert open(a).read() == open(b).read() and len(open(a).read().splitlines()) == 11


def test_estimated_thresholds_match_reference_values(oracle_bin, tmp_path):
    # thresholds the reference derives for its goldens (SURVEY.md verification note): f32-exact
    err = run_oracle(oracle_bin, fixture(BC), str(tmp_path / "o.bed"),
                     ["-i", "25", "-f", "1.0", "-p", "0.25", "--include-unmapped"])
    assert "threshold C 0.662109375 (n=109)" in err


def test_pileup_old_tags(oracle_bin, tmp_path):
    # tests/test_pileup.rs:161-192 — HG002_small (PacBio, old-style `Mm`/`Ml`, =/X CIGARs) through `update-tags --mode ambiguous
    # --no-implicit-probs` (restated test-side), then `pileup --no-filtering --only-tabs`: the reference's regression golden
    bam = update_tags_ambiguous(fixture("HG002_small.ch20._other.sorted.bam"), str(tmp_path / "updated.bam"))
    out = str(tmp_path / "out.bed")
    run_oracle(oracle_bin, bam, out, ["--no-filtering", "--only-tabs"])
    assert open(out).read() == open(fixture("pileup-old-tags-regressiontest.methyl.bed")).read()


@pytest.mark.parametrize("to_code", ["76792", "c"])
def test_pileup_chebi_code_same_output(oracle_bin, tmp_path, to_code):
    # tests/test_pileup.rs:373-444 — 5hmC renamed to a ChEBI number (and to another letter) by `adjust-mods --convert` (restated
    # test-side): the pileup must be the no-filter golden with the code renamed.  Pins ChEBI codes through tags, tallies and rows.
    from pileup_cases import convert_mod_code, chebi_case_expected_rows
    bam = convert_mod_code(fixture(BC), str(tmp_path / "conv.bam"), "h", to_code)
    out = str(tmp_path / "out.bed")
    run_oracle(oracle_bin, bam, out, ["-i", "25", "--no-filtering", "--only-tabs"])
    want, key = chebi_case_expected_rows(fixture("modbam.modpileup_nofilt.methyl.bed"), to_code)
    got = ["\t".join(f) for f in sorted((l.split("\t") for l in open(out).read().splitlines()), key=key)]
    assert got == want


def test_pileup_collapse(oracle_bin, tmp_path):
    # tests/test_pileup.rs:91-141 — `pileup --ignore h` on the fixture == `adjust-mods --ignore h` (restated test-side: the collapse
    # written back into the tags, ML re-quantised) followed by a plain pileup; -i 25 so that the interval chunking is exercised
    from pileup_cases import collapse_ignore
    bam = collapse_ignore(fixture(BC), str(tmp_path / "collapsed.bam"), "h")
    a, b = str(tmp_path / "collapsed.bed"), str(tmp_path / "restricted.bed")
    run_oracle(oracle_bin, bam, a, ["-i", "25", "--no-filtering"])
    run_oracle(oracle_bin, fixture(BC), b, ["-i", "25", "--ignore", "h", "--no-filtering"])
    assert open(a).read() and open(a).read() == open(b).read()


# ---- pileup-hemi (tests/test_pileup_hemi.rs)
@pytest.fixture(scope="module")
def hemi_ref(tmp_path_factory):
    return hemi_reference_fasta(tmp_path_factory.mktemp("hemi_ref"))


def run_oracle_hemi(oracle_bin, bam, out, flags, ok=True):
    p = subprocess.run([oracle_bin, "pileup-hemi", bam, "-o", out] + flags, capture_output=True, text=True)
    assert (p.returncode == 0) == ok, p.stderr
    return p.stderr


@pytest.mark.parametrize("name,flags,bam,golden", HEMI_GOLDEN_CASES, ids=[c[0] for c in HEMI_GOLDEN_CASES])
def test_oracle_reproduces_reference_hemi_golden(oracle_bin, tmp_path, hemi_ref, name, flags, bam, golden):
    out = str(tmp_path / "out.bed")
    run_oracle_hemi(oracle_bin, fixture(bam), out, flags + ["-r", hemi_ref])
    assert open(out).read() == open(fixture(golden)).read()


def test_hemi_combine_mods_folds_patterns(oracle_bin, tmp_path, hemi_ref):
    # DuplexModCall::into_combined (src/mod_bam.rs:1797-1829): every modified element becomes the base's any-mod code; the
    # per-position pattern total is unchanged
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    common = ["--cpg", "--region", "chr20:22,613,835-22,640,468", "--no-filtering", "-r", hemi_ref]
    run_oracle_hemi(oracle_bin, fixture(HEMI_BAM), a, common)
    run_oracle_hemi(oracle_bin, fixture(HEMI_BAM), b, common + ["--combine-mods"])
    def by_pos(path):
        d = {}
        for ln in open(path):
            f = ln.split("\t")
            d.setdefault(int(f[1]), []).append((f[3], int(f[11])))
        return d
    pa, pb = by_pos(a), by_pos(b)
    assert pa.keys() == pb.keys() and len(pa) == 191
    fold = lambda e: "-" if e == "-" else "C"
    for pos, rows in pa.items():
        want = {}
        for name, cnt in rows:
            x, y, base = name.split(",")
            k = "%s,%s,%s" % (fold(x), fold(y), base)
            want[k] = want.get(k, 0) + cnt
        assert dict(pb[pos]) == want


def test_hemi_requires_a_palindromic_motif(oracle_bin, tmp_path, hemi_ref):
    err = run_oracle_hemi(oracle_bin, fixture(HEMI_BAM), str(tmp_path / "o.bed"), ["--motif", "CGT", "0", "-r", hemi_ref], ok=False)
    assert "palindromic" in err
    err = run_oracle_hemi(oracle_bin, fixture(HEMI_BAM), str(tmp_path / "o.bed"), ["-r", hemi_ref], ok=False)
    assert "--cpg or a --motif" in err


# ---- summary (tests/test_summary.rs): the reference's tests assert on the ModSummary struct, not on text
def run_oracle_summary(oracle_bin, bam, flags):
    p = subprocess.run([oracle_bin, "summary", bam] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = {"rows": {}, "reads_with": {}, "threshold": {}}
    for ln in p.stdout.splitlines():
        f = ln.split("\t")
        if f[0] == "total_reads_used":
            out["total"] = int(f[1])
        elif f[0] == "reads_with":
            out["reads_with"][f[1]] = int(f[2])
        elif f[0] == "threshold":
            out["threshold"][f[1]] = float(f[2])
        elif f[0] == "row":
            out["rows"][(f[1], f[2])] = (int(f[3]), int(f[4]))
    return out


def test_summary_implicit_calls(oracle_bin):
    # tests/test_summary.rs:133-172: passthrough caller, BED of 8 positions on an implicit-mode `A+a.` read -> 8 canonical A calls
    s = run_oracle_summary(oracle_bin, fixture("single_read.bam"), ["--include-bed", fixture("include_bed_summary_test.bed"), "--no-filtering", "-i", "32", "--no-sampling"])
    assert s["total"] == 1 and s["reads_with"] == {"A": 1}
    assert s["rows"][("A", "-")] == (8, 0) and all(v == (0, 0) for k, v in s["rows"].items() if k != ("A", "-"))


def test_summary_ignore(oracle_bin):
    # tests/test_summary.rs:31-68: the states counted without / with ReDistribute('h')
    a = run_oracle_summary(oracle_bin, fixture(BC), ["-i", "25", "--no-sampling"])
    b = run_oracle_summary(oracle_bin, fixture(BC), ["-i", "25", "--no-sampling", "--ignore", "h"])
    assert {k for k, v in a["rows"].items() if v[0]} == {("C", "-"), ("C", "h"), ("C", "m")}
    assert {k for k, v in b["rows"].items() if v[0]} == {("C", "-"), ("C", "m")}


def test_summary_edge_filter(oracle_bin):
    # tests/test_summary.rs:70-131 (the part that needs no adjust-mods): same reads, fewer calls
    a = run_oracle_summary(oracle_bin, fixture(BC), ["-i", "25", "--no-sampling"])
    b = run_oracle_summary(oracle_bin, fixture(BC), ["-i", "25", "--no-sampling", "--edge-filter", "50"])
    assert a["reads_with"]["C"] == b["reads_with"]["C"] and a["total"] == b["total"]
    assert sum(v[0] for v in a["rows"].values()) > sum(v[0] for v in b["rows"].values())
