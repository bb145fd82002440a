"""BAMs for the record sampler's corners (reads_sampler/mod.rs): a tail of records without coordinates, with or without an index."""
import random

from bamfuzz import aux_bc, aux_z, bam_header, bam_record, bgzf_write, write_bai


def unmapped_tail_bam(prefix, n_mapped=30, n_unmapped=600, seed=11, index=True, contig_len=6000):
    """A few mapped reads (fewer than 100 sampled records: reads_sampler/mod.rs:89-90 then always turns to the unmapped ones) and a tail of
    records without coordinates, each with a `C+m?` tag; every tenth record carries no tags (the iterator never offers it to the sampler)."""
    r = random.Random(seed)
    contigs = [("ctg", contig_len)]
    data = bytearray(bam_header(contigs))
    idx = []

    def one(tid, pos, flag, k):
        n = r.randrange(60, 220)
        seq = "".join(r.choice("ACGT") for _ in range(n))
        ncalls = min(seq.count("C"), r.randrange(1, 12))
        aux = b"" if (k % 10 == 9 or ncalls == 0) else aux_z("MM", "C+m?," + ",".join("0" for _ in range(ncalls)) + ";") + aux_bc("ML", [r.randrange(256) for _ in range(ncalls)])
        cigar = [(n, "M")] if tid >= 0 else []
        rec = bam_record(tid, pos, flag, "r%05d" % k, cigar, seq, aux)
        idx.append((tid, pos, n if tid >= 0 else 0, flag, len(data), len(rec)))
        data.extend(rec)

    starts = sorted(r.randrange(0, contig_len - 1000) for _ in range(n_mapped))
    for k, s in enumerate(starts):
        one(0, s, 0, k)
    for k in range(n_unmapped):
        one(-1, -1, 4, n_mapped + k)
    offs = bgzf_write(prefix + ".bam", bytes(data))
    if index:
        write_bai(prefix + ".bam.bai", 1, offs, idx)
    return prefix + ".bam"
