"""The oracle's sampler on a BAM without an index (reads_sampler/mod.rs:129-158): no schedule, the file in file order under RecordSampler —
`-n N` = the first N records that yield values (record_sampler.rs:72-78, 97-99).  Pinned by construction: sampling N of a file == sampling
all of the file cut behind its N-th usable record."""
import subprocess

from sampler_cases import unmapped_tail_bam


def table(oracle_bin, bam, flags):
    p = subprocess.run([oracle_bin, "sample-probs", bam, "-p", "0.1,0.5,0.9"] + flags, capture_output=True, text=True)
    return p.stdout if p.returncode == 0 else None


def test_first_n_records_of_an_unindexed_file(oracle_bin, tmp_path):
    # 260 mapped + 200 unmapped records; every tenth record has no tags: of the first 50 records 45 yield values
    full = unmapped_tail_bam(str(tmp_path / "full"), n_mapped=260, n_unmapped=200, index=False, contig_len=40000)
    head = str(tmp_path / "head.bam")   # the same file cut behind its 50th record
    import gzip
    import struct
    raw = gzip.open(full).read()
    off = 4 + 4 + struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, off)[0]; off += 4
    for _ in range(n_ref):
        off += 4 + struct.unpack_from("<i", raw, off)[0] + 4
    end = off
    for _ in range(50):
        end += 4 + struct.unpack_from("<i", raw, end)[0]
    from bamfuzz import bgzf_write
    bgzf_write(head, raw[:end])
    assert table(oracle_bin, full, ["-n", "45"]) == table(oracle_bin, head, ["--no-sampling"])
    assert table(oracle_bin, full, ["-n", "45"]) != table(oracle_bin, full, ["-n", "46"])
    assert table(oracle_bin, full, ["-n", "10042"]) == table(oracle_bin, full, ["--no-sampling"])
    assert table(oracle_bin, full, ["--region", "ctg"]) is None                      # "cannot use region without indexed BAM"
    assert table(oracle_bin, full, ["-f", "0.5"]) is None                            # entropy-seeded in the reference
    a, b = table(oracle_bin, full, ["-f", "0.5", "--seed", "1"]), table(oracle_bin, full, ["-f", "0.5", "--seed", "2"])
    assert a is not None and b is not None and a != b and a == table(oracle_bin, full, ["-f", "0.5", "--seed", "1"])
