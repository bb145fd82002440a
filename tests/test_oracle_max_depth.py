"""The oracle's restatement of htslib's maxcnt rule, pinned on hand-worked piles (htslib is a dependency of rust-htslib, not in the
reference tree: the rule is read from its published bam_plp_push — parity unpinned by any reference fixture).  Expected counts below
follow from: the first record of a start is always buffered; a later record of the same start is refused when
1 + #(buffered records ending at or behind that start) > cap."""
import subprocess

from max_depth_cases import PAIRS, SAME_START, STAGGERED, TWO_STACKS, coverage_by_position, pile


def run(oracle_bin, tmp_path, bam, flags):
    out = str(tmp_path / "o.bed")
    p = subprocess.run([oracle_bin, "pileup", bam, out, "--no-filtering"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-300:]
    return coverage_by_position(open(out).read())


def test_same_start_pile_is_cut_at_the_cap(oracle_bin, tmp_path):
    bam = pile(str(tmp_path / "a"), SAME_START)
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "10"]) == {101: 10}     # records 1..10: 1 + (k - 1) <= 10
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "29"]) == {101: 29}
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "30"]) == {101: 30}
    assert run(oracle_bin, tmp_path, bam, []) == {101: 30}


def test_distinct_starts_are_never_dropped(oracle_bin, tmp_path):
    bam = pile(str(tmp_path / "b"), STAGGERED)
    want = {101 + i: 1 for i in range(30)}
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "10"]) == want
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "1", "-i", "13"]) == want


def test_pairs_lose_their_second_read_over_the_cap(oracle_bin, tmp_path):
    bam = pile(str(tmp_path / "c"), PAIRS)
    cov = run(oracle_bin, tmp_path, bam, ["--max-depth", "10"])
    # pair p (starts 100 + 2p): its second read sees 1 + (buffered so far) = 1 + (2p + 1) while both reads of every earlier pair were taken,
    # i.e. p <= 4; from p = 5 on the buffer holds 11, 12, ... and every second read is refused
    assert cov == {101 + 2 * p: (2 if p <= 4 else 1) for p in range(15)}


def test_a_fetch_starts_a_fresh_iterator(oracle_bin, tmp_path):
    bam = pile(str(tmp_path / "d"), TWO_STACKS)
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "20"]) == {101: 12, 131: 8}    # record k of the second stack: 1 + 12 + (k - 1) <= 20 -> k <= 8
    # intervals of 25 bases: [125, 150) is fetched by itself — all 24 records overlap it, the rule is the same
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "20", "-i", "25"]) == {101: 12, 131: 8}
    # with the first stack gone from the fetch ([175, 200) holds only the second stack's tails) nothing is over the cap: no call there anyway
    assert run(oracle_bin, tmp_path, bam, ["--max-depth", "24"]) == {101: 12, 131: 12}
