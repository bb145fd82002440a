"""Piles for htslib's maxcnt rule (bam_plp_push: a record is refused when it starts where the last buffered record started and the buffer
holds more than --max-depth entries; pileup/mod.rs:755-759 sets the cap).  Every read is 60 bases with one `C+m?` call on its first C."""
import random

from bamfuzz import aux_bc, aux_z, bam_header, bam_record, bgzf_write, write_bai

CONTIG = ("ctg", 2000)


def pile(prefix, starts, length=60, index=False, seed=3):
    r = random.Random(seed)
    data = bytearray(bam_header([CONTIG]))
    idx = []
    for k, s in enumerate(starts):
        seq = "AC" + "".join(r.choice("AGT") for _ in range(length - 2))   # one C, at read position 1 -> the call sits at s + 1
        aux = aux_z("MM", "C+m?,0;") + aux_bc("ML", [250])
        rec = bam_record(0, s, 0, "r%04d" % k, [(length, "M")], seq, aux)
        idx.append((0, s, length, 0, len(data), len(rec)))
        data.extend(rec)
    offs = bgzf_write(prefix + ".bam", bytes(data))
    if index:
        write_bai(prefix + ".bam.bai", 1, offs, idx)
    return prefix + ".bam"


def coverage_by_position(bed_text):
    """bedMethyl rows -> {position: valid coverage (column 10)}"""
    out = {}
    for ln in bed_text.splitlines():
        f = ln.split("\t")
        out[int(f[1])] = int(f[9])
    return out


SAME_START = [100] * 30                          # one start: the first is always buffered, then reads while the buffer holds <= cap
STAGGERED = [100 + i for i in range(30)]         # thirty different starts: nothing is ever refused, whatever the cap
PAIRS = [100 + 2 * (i // 2) for i in range(30)]  # two reads per start: the second of a pair is refused once the buffer is over the cap
TWO_STACKS = [100] * 12 + [130] * 12             # the second stack meets the first one's twelve buffered reads
