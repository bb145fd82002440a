"""modkit_amd.read_bedmethyl / rows_digest (what bench.py and the re-launch parity tests use to compare row columns fetched from the device
with a bedMethyl FILE): on the reference's own golden files the reader returns one row per line, the columns in their places, and the
invariants PileupFeatureCounts guarantees (pileup/mod.rs:283-410)."""
import os

import numpy as np
import pytest

import modkit_amd

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modkit_fixtures")


@pytest.mark.parametrize("name", ["modbam.modpileup_nofilt.methyl.bed", "modbam.modpileup_combined.methyl.bed", "bc_anchored_10_reads_nofilt_cg_motif_strand_combine.bed",
                                  "pileup-old-tags-regressiontest.methyl.bed"])
def test_reader_on_reference_goldens(name):
    path = os.path.join(FIX, name)
    lines = [l.split() for l in open(path) if l.strip()]
    r = modkit_amd.read_bedmethyl(path)
    assert len(r["pos"]) == len(lines) > 0
    k = len(lines) // 2
    assert int(r["pos"][k]) == int(lines[k][1]) and chr(int(r["strand"][k])) == lines[k][5] and chr(int(r["code_repr"][k])) == lines[k][3]
    assert [int(r[f][k]) for f in ("n_valid", "n_mod", "n_canonical", "n_other", "n_delete", "n_fail", "n_diff", "n_nocall")] == [int(x) for x in (lines[k][9], *lines[k][11:18])]
    assert np.array_equal(r["n_valid"], r["n_mod"] + r["n_canonical"] + r["n_other"])
    d = modkit_amd.rows_digest(r)
    r["n_nocall"] = r["n_nocall"].copy(); r["n_nocall"][k] += 1
    assert modkit_amd.rows_digest(r) != d


def test_chebi_codes_and_empty_file(tmp_path):
    p = tmp_path / "x.bed"
    p.write_text("chr1\t5\t6\t76792\t3\t+\t5\t6\t255,0,0\t3\t33.33\t1\t2\t0\t0\t0\t0\t1\n")
    r = modkit_amd.read_bedmethyl(str(p))
    assert int(r["code_repr"][0]) == (0x80000000 | 76792) and int(r["n_nocall"][0]) == 1
    e = tmp_path / "e.bed"; e.write_text("")
    assert len(modkit_amd.read_bedmethyl(str(e))["pos"]) == 0
