"""GPU: `--max-depth` (htslib's maxcnt, pileup/mod.rs:755-759).  The device path does not drop records per (record, interval); it decides
exactly whether htslib WOULD drop any (depth_guard, mkp_api.cpp) — if not, it runs and equals the oracle at any depth; if so, it refuses
loudly.  Rounds 1-5 refused every column deeper than the cap, although htslib only ever refuses records that share their start with the
record buffered last."""
import subprocess

import pytest

import modkit_amd
from max_depth_cases import PAIRS, SAME_START, STAGGERED, TWO_STACKS, pile

pytestmark = pytest.mark.gpu


def both(oracle_bin, tmp_path, bam, flags):
    dev, ora = str(tmp_path / "dev.bed"), str(tmp_path / "ora.bed")
    modkit_amd.pileup([bam, dev, "--no-filtering"] + flags)
    p = subprocess.run([oracle_bin, "pileup", bam, ora, "--no-filtering"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-300:]
    a, b = open(dev).read(), open(ora).read()
    assert a == b and a
    return a


@pytest.mark.parametrize("index", [False, True])
def test_deeper_than_the_cap_without_shared_starts_runs(oracle_bin, tmp_path, index):
    bam = pile(str(tmp_path / "s"), STAGGERED, index=index)
    for flags in (["--max-depth", "10"], ["--max-depth", "1", "-i", "13"], ["--max-depth", "29"]):
        assert len(both(oracle_bin, tmp_path, bam, flags).splitlines()) == 30


@pytest.mark.parametrize("index", [False, True])
def test_stacks_under_the_cap_run_and_over_it_are_refused(oracle_bin, tmp_path, index):
    bam = pile(str(tmp_path / "t"), TWO_STACKS, index=index)
    both(oracle_bin, tmp_path, bam, ["--max-depth", "24"])            # 1 + 12 + 11 = 24: the last record of the second stack still fits
    both(oracle_bin, tmp_path, bam, ["--max-depth", "24", "-i", "25"])
    for flags in (["--max-depth", "23"], ["--max-depth", "20", "-i", "25"], ["--max-depth", "11"]):
        with pytest.raises(modkit_amd.MkpError) as e:
            modkit_amd.pileup([bam, str(tmp_path / "x.bed"), "--no-filtering"] + flags)
        assert e.value.status == -3 and "max_depth" in str(e.value)


def test_same_start_piles_and_pairs(oracle_bin, tmp_path):
    a = pile(str(tmp_path / "a"), SAME_START)
    both(oracle_bin, tmp_path, a, ["--max-depth", "30"])
    both(oracle_bin, tmp_path, a, [])
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup([a, str(tmp_path / "x.bed"), "--no-filtering", "--max-depth", "29"])
    c = pile(str(tmp_path / "c"), PAIRS)
    both(oracle_bin, tmp_path, c, ["--max-depth", "30"])
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup([c, str(tmp_path / "x.bed"), "--no-filtering", "--max-depth", "10"])
    # a stack that straddles an interval start: in the second interval's fetch its records start in front of the interval
    both(oracle_bin, tmp_path, a, ["--max-depth", "30", "-i", "120"])
    with pytest.raises(modkit_amd.MkpError):
        modkit_amd.pileup([a, str(tmp_path / "x.bed"), "--no-filtering", "--max-depth", "29", "-i", "120"])
