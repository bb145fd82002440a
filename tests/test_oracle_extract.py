"""Pins the oracle's restatement of `modkit extract calls` (oracle/oracle_extract.hpp) on the reference's own golden tables:
tests/test_extract.rs:499-521 (test_extract_calls_regression: estimated thresholds, --ref k-mers) and :523-560
(test_extract_supplementary_secondary: --allow-non-primary, only within-alignment calls of a secondary record).  Whole-file byte
comparison, as check_against_expected_text_file does.  Plus unit known answers of the pieces the table is made of."""
import subprocess

import pytest

from pileup_cases import EXTRACT_CALLS_CASES, REF, fixture


@pytest.mark.parametrize("name,flags,bam,golden", EXTRACT_CALLS_CASES, ids=[c[0] for c in EXTRACT_CALLS_CASES])
def test_oracle_reproduces_extract_calls_golden(oracle_bin, tmp_path, name, flags, bam, golden):
    out = str(tmp_path / "calls.tsv")
    p = subprocess.run([oracle_bin, "extract-calls", fixture(bam), out] + [f.format(ref=REF) for f in flags], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(out).read() == open(fixture(golden)).read()


def test_pass_only_and_no_filtering(oracle_bin, tmp_path):
    # --pass-only drops exactly the rows whose `fail` column is true; --no-filtering fails nothing
    a, b, c = (str(tmp_path / n) for n in ("a.tsv", "b.tsv", "c.tsv"))
    bam = fixture("2_reads_all_context.bam")
    for out, fl in ((a, []), (b, ["--pass-only"]), (c, ["--no-filtering"])):
        p = subprocess.run([oracle_bin, "extract-calls", bam, out] + fl, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
    rows = [l.split("\t") for l in open(a).read().splitlines()[1:]]
    kept = ["\t".join(r) for r in rows if r[17] == "false"]
    assert kept and open(b).read().splitlines()[1:] == kept
    assert all(l.split("\t")[17] == "false" for l in open(c).read().splitlines()[1:])
