"""The oracle's `extract calls --num-reads` on an indexed BAM, pinned on a hand-worked schedule (no reference fixture holds one): thirty
reads on one 2 000-base contig, all starting in [100, 130), one call each.  SamplingSchedule::from_num_reads gives the contig
min(ceil(N * 1.0), 30) reads; an interval's sampler takes ceil(that * interval length / length of its super batch) records
(get_record_sampler, sampling_schedule.rs:417-438), and a super batch is floor(threads * 1.5) groups of >= --interval-size bases."""
import subprocess

from max_depth_cases import STAGGERED, pile


def reads_of(oracle_bin, tmp_path, bam, flags):
    out = str(tmp_path / "o.tsv")
    p = subprocess.run([oracle_bin, "extract-calls", bam, out, "--no-filtering"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-300:]
    return [ln.split("\t")[0] for ln in open(out).read().splitlines()[1:]]


def test_quota_per_interval_follows_the_super_batch(oracle_bin, tmp_path):
    bam = pile(str(tmp_path / "s"), STAGGERED, index=True)
    first = lambda n: ["r%04d" % k for k in range(n)]
    # -t 1: one group per super batch -> the interval [0, 500) is its whole batch: ceil(10 * 500 / 500) = 10 of the reads that start in it
    assert reads_of(oracle_bin, tmp_path, bam, ["--num-reads", "10", "-i", "500", "-t", "1"]) == first(10)
    # -t 4: six groups per super batch -> all four intervals share one: ceil(10 * 500 / 2000) = 3
    assert reads_of(oracle_bin, tmp_path, bam, ["--num-reads", "10", "-i", "500", "-t", "4"]) == first(3)
    # intervals of 110 bases, -t 1: [0, 110) takes ceil(10 * 110 / 110) = 10 of its ten reads (starts 100..109), [110, 220) ten of its twenty
    assert reads_of(oracle_bin, tmp_path, bam, ["--num-reads", "10", "-i", "110", "-t", "1"]) == first(20)
    # N beyond the index count: the contig's quota is capped at its 30 reads; one interval of the whole contig takes them all
    assert reads_of(oracle_bin, tmp_path, bam, ["--num-reads", "500", "-i", "2000", "-t", "1"]) == first(30)
    # the serial scan's N is plain: the first N records
    assert reads_of(oracle_bin, tmp_path, bam, ["--num-reads", "7", "--ignore-index"]) == first(7)
