"""Seeded modBAM fuzzer for the differential tests (oracle vs device): writes a coordinate-sorted BGZF BAM
plus FASTA and optional BED with the structures the reference's decoder distinguishes — both strands,
soft/hard clips, insertions, deletions, ref-skips, =/X ops, `?` / `.` / no-mode tags, combined (`C+hm?`)
and split (`C+h?;C+m?`) tags with equal or different position lists, `N+x` tags, negative-strand
(`G-m`) tags, ChEBI codes, probability ties, broken tags (short ML, bad MN, runaway delta lists,
missing ML), secondary / supplementary / duplicate / QC-fail records and reads with no tags."""
import random
import struct
import zlib

NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def bgzf_write(path, data):
    """-> file offset of every BGZF block (each holds 0xff00 inflated bytes, the last one fewer)"""
    offs = []
    with open(path, "wb") as f:
        for off in range(0, len(data), 0xff00):
            chunk = data[off:off + 0xff00]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            bsize = len(comp) + 25
            offs.append(f.tell())
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize))
            f.write(comp)
            f.write(struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        offs.append(f.tell())
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return offs


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write_bai(path, n_ref, block_offs, records):
    """BAI (SAM spec 5.2) as `samtools index` writes it: bins with chunk lists, the 16 kb linear index, htslib's metadata pseudo-bin 37450
    (file range + mapped / unmapped counts) and the count of records without coordinates.
    records = (tid, pos, reflen, flag, inflated offset of the record, its inflated length), file order."""
    def voff(u):
        return (block_offs[u // 0xff00] << 16) | (u % 0xff00)
    refs = [dict(bins={}, lin=[], beg=None, end=0, mapped=0, unmapped=0) for _ in range(n_ref)]
    no_coor = 0
    for tid, pos, reflen, flag, u0, ulen in records:
        if tid < 0:
            no_coor += 1
            continue
        R = refs[tid]
        vb, ve = voff(u0), voff(u0 + ulen)
        end = pos + (reflen if reflen > 0 else 1)
        chunks = R["bins"].setdefault(reg2bin(pos, end), [])
        if chunks and chunks[-1][1] == vb:
            chunks[-1][1] = ve
        else:
            chunks.append([vb, ve])
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            while len(R["lin"]) <= w:
                R["lin"].append(0)
            if R["lin"][w] == 0:
                R["lin"][w] = vb
        if R["beg"] is None:
            R["beg"] = vb
        R["end"] = ve
        if flag & 4:
            R["unmapped"] += 1
        else:
            R["mapped"] += 1
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for R in refs:
        for i in range(1, len(R["lin"])):
            if R["lin"][i] == 0:
                R["lin"][i] = R["lin"][i - 1]
        has = R["beg"] is not None
        out += struct.pack("<i", len(R["bins"]) + (1 if has else 0))
        for b in sorted(R["bins"]):
            out += struct.pack("<Ii", b, len(R["bins"][b]))
            for vb, ve in R["bins"][b]:
                out += struct.pack("<QQ", vb, ve)
        if has:
            out += struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", R["beg"], R["end"], R["mapped"], R["unmapped"])
        out += struct.pack("<i", len(R["lin"])) + b"".join(struct.pack("<Q", v) for v in R["lin"])
    out += struct.pack("<Q", no_coor)
    with open(path, "wb") as f:
        f.write(bytes(out))


def bam_record(tid, pos, flag, qname, cigar, seq, aux):
    qn = qname.encode() + b"\0"
    cg = b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(op)) for l, op in cigar)
    sq = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        sq[i // 2] |= NT16[c] << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), 60, 4680, len(cigar), flag, len(seq), -1, -1, 0) + qn + cg + bytes(sq) + b"\xff" * len(seq) + aux
    return struct.pack("<i", len(body)) + body


def aux_z(tag, s):
    return tag.encode() + b"Z" + s.encode() + b"\0"


def aux_bc(tag, vals):
    return tag.encode() + b"BC" + struct.pack("<I", len(vals)) + bytes(vals)


def aux_i(tag, v):
    return tag.encode() + b"i" + struct.pack("<i", v)


def bam_header(contigs):
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    data = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs)))
    for name, ln in contigs:
        data += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    return data


class Fuzz:
    def __init__(self, seed, contigs=(("ctgA", 12000), ("ctgB", 3000)), n_reads=250, mean_len=900, profile="mixed", tie_rate=0.05, weird_rate=0.08, index=None, dup_rate=0.0):
        self.dup_rate = dup_rate   # share of reads written as TWO primary records with one name: an unmarked duplicate, a shifted copy, or the two halves of a split read
        self.index = (seed % 3 != 2) if index is None else index   # two of three seeds write a BAI
        self.r = random.Random(seed)
        self.contigs = contigs
        self.n_reads = n_reads
        self.mean_len = mean_len
        self.profile = profile
        self.tie_rate = tie_rate
        self.weird_rate = weird_rate
        self.refs = {}
        for name, ln in contigs:
            s = [self.r.choice("ACGT") for _ in range(ln)]
            for _ in range(ln // 40):  # extra CpGs
                p = self.r.randrange(ln - 1)
                s[p], s[p + 1] = "C", "G"
            self.refs[name] = "".join(s)

    # ---- MM/ML construction on the as-sequenced read
    def tags_for(self, fwd):
        r = self.r
        prof = self.profile
        if prof == "mixed":
            prof = r.choice(["m", "hm_comb", "hm_split", "hm_split_diff", "hma", "implicit", "default", "duplex", "nbase", "chebi", "m"])
        mm, ml = [], []

        def positions(base, keep=1.0, cpg=False):
            occ = [i for i, c in enumerate(fwd) if c == base]
            if cpg:
                sel = [k for k, i in enumerate(occ) if i + 1 < len(fwd) and fwd[i + 1] == "G"]
            else:
                sel = list(range(len(occ)))
            return [k for k in sel if r.random() < keep]  # ranks among occurrences of `base`

        def deltas(ranks):
            out, last = [], -1
            for k in ranks:
                out.append(k - last - 1)
                last = k
            return out

        def q():
            x = r.random()
            return r.randrange(0, 40) if x < 0.4 else r.randrange(215, 256) if x < 0.8 else r.randrange(256)

        def header(h, ranks):
            d = deltas(ranks)
            mm.append(h + ("," + ",".join(map(str, d)) if d else "") + ";")

        def add(h, ranks, ncodes):
            header(h, ranks)
            for _ in ranks:
                if ncodes == 1:
                    ml.append(q())
                elif r.random() < self.tie_rate:
                    ml.extend([r.randrange(0, 120)] * ncodes)
                else:
                    a = q()
                    rest = max(0, 255 - a)
                    vals = [a] + [r.randrange(0, rest // (ncodes - 1) + 1) for _ in range(ncodes - 1)]
                    r.shuffle(vals)
                    ml.extend(vals)

        if prof == "m":
            add("C+m?", positions("C", 0.95, cpg=True), 1)
        elif prof == "hm_comb":
            add("C+hm?", positions("C", 0.95, cpg=True), 2)
        elif prof == "hm_split":
            rk = positions("C", 0.95, cpg=True)
            header("C+h?", rk)
            header("C+m?", rk)
            hv, mv = [], []
            for _ in rk:
                if r.random() < self.tie_rate:
                    a = r.randrange(0, 120)
                    hv.append(a)
                    mv.append(a)
                else:
                    a = q()
                    b = r.randrange(0, max(1, 256 - a))
                    if r.random() < 0.5:
                        a, b = b, a
                    hv.append(a)
                    mv.append(b)
            ml.extend(hv)
            ml.extend(mv)
        elif prof == "hm_split_diff":
            rk1 = positions("C", 0.5)
            header("C+h?", rk1)
            ml.extend(r.randrange(0, 128) if r.random() < 0.97 else r.randrange(128, 256) for _ in rk1)
            rk2 = positions("C", 0.5)
            header("C+m?", rk2)
            ml.extend(r.randrange(0, 128) if r.random() < 0.97 else r.randrange(128, 256) for _ in rk2)
        elif prof == "hma":
            rk = positions("C", 0.9, cpg=True)
            header("C+h?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            header("C+m?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            add("A+a?", positions("A", 0.97), 1)
        elif prof == "implicit":
            add("C+m.", positions("C", 0.3), 1)
        elif prof == "default":
            add("C+m", positions("C", 0.3), 1)
        elif prof == "duplex":
            add("C+m?", positions("C", 0.9, cpg=True), 1)
            add("G-m?", positions("G", 0.5), 1)
        elif prof == "duplex_hm":   # both strands of a duplex read, two codes each (pileup-hemi patterns like h,m / m,- / -,-)
            add("C+hm?", positions("C", 0.9, cpg=True), 2)
            add("G-hm?", positions("G", 0.7), 2)
        elif prof == "duplex_split":  # the same calls as separate tags per code, plus 6mA on one strand
            rk = positions("C", 0.9, cpg=True)
            header("C+h?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            header("C+m?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            add("G-m?", positions("G", 0.8), 1)
            add("A+a?", positions("A", 0.5), 1)
        elif prof == "duplex_chebi":  # a ChEBI-numbered code next to a letter code on both strands (pattern names like 76792,m,C)
            rk = positions("C", 0.9, cpg=True)
            header("C+76792?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            header("C+m?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            rg = positions("G", 0.7)
            header("G-76792?", rg)
            ml.extend(r.randrange(0, 128) for _ in rg)
            header("G-m?", rg)
            ml.extend(r.randrange(0, 128) for _ in rg)
        elif prof == "duplex_3codes":  # three codes per strand: 16 patterns per base
            add("C+hmf?", positions("C", 0.9, cpg=True), 3)
            add("G-hmf?", positions("G", 0.7), 3)
        elif prof == "nbase":
            n = len(fwd)
            pos = sorted(r.sample(range(n), min(n, r.randrange(0, 25))))
            d, last = [], -1
            for p in pos:
                d.append(p - last - 1)
                last = p
            mm.append("N+b?" + ("," + ",".join(map(str, d)) if d else "") + ";")
            ml.extend(q() for _ in pos)
            add("C+m?", positions("C", 0.9, cpg=True), 1)
        elif prof == "chebi":
            rk = positions("C", 0.9, cpg=True)
            header("C+76792?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
            header("C+m?", rk)
            ml.extend(r.randrange(0, 128) for _ in rk)
        return "".join(mm), ml

    def make_read(self, ref):
        r = self.r
        L = max(30, min(len(ref) - 10, int(r.lognormvariate(0, 0.5) * self.mean_len)))
        start = r.randrange(0, max(1, len(ref) - L))
        reverse = r.random() < 0.5
        cigar, seq = [], []

        def push(n, op):
            if n <= 0:
                return
            if cigar and cigar[-1][1] == op:
                cigar[-1][0] += n
            else:
                cigar.append([n, op])

        if r.random() < 0.1:
            push(r.randrange(1, 20), "H")
        sc = r.randrange(0, 25) if r.random() < 0.6 else 0
        push(sc, "S")
        seq.extend(r.choice("ACGT") for _ in range(sc))
        p = start
        use_eqx = r.random() < 0.15
        first = True
        while p < start + L and p < len(ref):
            x = r.random()
            c = ref[p].upper()
            if first:
                x = 1.0  # alignments start on a match
                first = False
            if x < 0.015:
                n = r.randrange(1, 5)
                push(n, "I")
                seq.extend(r.choice("ACGT") for _ in range(n))
            elif x < 0.03:
                n = r.randrange(1, 6)
                push(n, "D")
                p += n
            elif x < 0.033:
                n = r.randrange(5, 60)
                push(n, "N")
                p += n
            elif x < 0.05:
                b = r.choice([k for k in "ACGT" if k != c] + (["N"] if r.random() < 0.05 else []))
                push(1, "X" if use_eqx else "M")
                seq.append(b)
                p += 1
            else:
                push(1, "=" if use_eqx else "M")
                seq.append(c if c in "ACGT" else "N")
                p += 1
        while cigar and cigar[-1][1] in "IDN":  # and end on one
            n, op = cigar.pop()
            if op == "I":
                del seq[len(seq) - n:]
        ec = r.randrange(0, 25) if r.random() < 0.6 else 0
        push(ec, "S")
        seq.extend(r.choice("ACGT") for _ in range(ec))
        seq = "".join(seq)
        if not any(op in "M=X" for _, op in cigar):
            return None
        fwd = revcomp(seq) if reverse else seq
        flag = 16 if reverse else 0
        x = r.random()
        if x < 0.02:
            flag |= 256
        elif x < 0.04:
            flag |= 1024
        elif x < 0.06:
            flag |= 2048
        elif x < 0.07:
            flag |= 512
        aux = b""
        if r.random() >= 0.04:  # else: no tags at all
            mm, ml = self.tags_for(fwd)
            kind = r.random()
            mmt, mlt = ("Mm", "Ml") if r.random() < 0.1 else ("MM", "ML")
            wr = self.weird_rate
            if kind < wr * 0.25:
                ml = ml[:max(0, len(ml) - r.randrange(1, 4))]            # ML too short
            elif kind < wr * 0.5:
                mm = mm.replace(";", ",%d;" % (len(fwd) + 5), 1)          # runs past the end of the read
                ml = ml + [7]
            elif kind < wr * 0.75:
                aux += aux_i("MN", len(seq) + 1)                          # MN mismatch
            elif kind < wr:
                mm = "Q" + mm[1:] if mm else "C*m?;"                      # invalid header
            if wr and kind > 0.97:
                aux += aux_z(mmt, mm)                                     # ML missing
            else:
                aux += aux_z(mmt, mm) + aux_bc(mlt, ml)
                if (flag & (256 | 2048) or r.random() < 0.1) and b"MN" not in aux:
                    aux += aux_i("MN", len(seq))
        return start, flag, [(n, op) for n, op in cigar], seq, aux

    def second_record(self, start, flag, cigar, seq, aux, name, ref_len):
        """A second PRIMARY record with the same name (the reference keys its per-interval read cache by name, read_cache.rs:28-35): an unmarked
        duplicate, the same alignment shifted by a few bases, or the read split in two (each half soft-clips the other's bases; the first
        record is replaced by the first half)."""
        kind = self.r.choice(["copy", "shift", "split"])
        span = sum(n for n, op in cigar if op in "MDN=X")
        if kind == "copy":
            return [(start, flag, cigar, seq, aux, name)]
        if kind == "shift":
            d = self.r.randrange(1, 40)
            return [(start + d, flag, cigar, seq, aux, name)] if start + d + span < ref_len else []
        # split: cut at a query position inside an M run
        qtot = sum(n for n, op in cigar if op in "MIS=X")
        if qtot < 40:
            return []
        cut = self.r.randrange(15, qtot - 15)
        a, b, q, r = [], [], 0, start
        b_start = None
        for n, op in cigar:
            qn = n if op in "MIS=X" else 0
            rn = n if op in "MDN=X" else 0
            if q + qn <= cut or (qn == 0 and q < cut):
                a.append((n, op)); q += qn; r += rn
            elif q >= cut:
                if b_start is None:
                    if op in "DN":      # a half must not begin with a deletion / skip
                        r += rn
                        continue
                    b_start = r
                b.append((n, op)); q += qn; r += rn
            else:                       # the cut falls inside this op
                k1 = cut - q
                if op in "M=X":
                    a.append((k1, op)); b_start = r + k1; b.append((n - k1, op))
                else:                   # inside an insertion / soft clip: give it whole to the first half
                    a.append((n, op)); cut = q + n
                q += qn; r += rn
        if b_start is None or not any(op in "M=X" for _, op in a) or not any(op in "M=X" for _, op in b):
            return []
        while a and a[-1][1] in "DN":
            a.pop()
        qa = sum(n for n, op in a if op in "MIS=X")
        a.append((qtot - qa, "S"))
        b = [(qa, "S")] + b
        return [("replace", a), (b_start, flag, b, seq, aux, name)]

    def write(self, prefix, bed=False):
        data = bam_header(self.contigs)
        k = 0
        index = []
        total = sum(c[1] for c in self.contigs)
        for tid, (name, ln) in enumerate(self.contigs):
            reads = []
            for _ in range(max(1, self.n_reads * ln // total)):
                rd = self.make_read(self.refs[name])
                if rd:
                    reads.append(rd)
            named = []
            for start, flag, cigar, seq, aux in reads:
                name = "read%06d" % k
                k += 1
                named.append((start, flag, cigar, seq, aux, name))
                if self.dup_rate and self.r.random() < self.dup_rate and not (flag & (4 | 256 | 2048)):
                    for extra in self.second_record(start, flag, cigar, seq, aux, name, ln):
                        if extra[0] == "replace":
                            named[-1] = (start, flag, extra[1], seq, aux, name)
                        else:
                            named.append(extra)
            named.sort(key=lambda t: t[0])   # (stable: the first record of a name stays in front of its copy at the same start)
            for start, flag, cigar, seq, aux, name in named:
                rec = bam_record(tid, start, flag, name, cigar, seq, aux)
                index.append((tid, start, sum(n for n, op in cigar if op in "MDN=X"), flag, len(data), len(rec)))
                data += rec
        offs = bgzf_write(prefix + ".bam", bytes(data))
        if self.index:   # an indexed BAM goes through the device ingest (linear-index entry points), an unindexed one through the host loader
            write_bai(prefix + ".bam.bai", len(self.contigs), offs, index)
        with open(prefix + ".fa", "w") as f:
            for name, _ in self.contigs:
                s = self.refs[name]
                s = s[:50] + s[50:120].lower() + s[120:]  # soft-masked stretch (--mask)
                f.write(">%s\n" % name)
                for i in range(0, len(s), 60):
                    f.write(s[i:i + 60] + "\n")
        if bed:
            with open(prefix + ".bed", "w") as f:
                for name, ln in self.contigs:
                    for _ in range(12):
                        a = self.r.randrange(0, ln - 50)
                        b = min(ln, a + self.r.randrange(10, 600))
                        if self.r.random() < 0.4:
                            f.write("%s\t%d\t%d\n" % (name, a, b))
                        else:
                            f.write("%s\t%d\t%d\tx\t0\t%s\n" % (name, a, b, self.r.choice("+-.")))
        return prefix + ".bam", prefix + ".fa", prefix + ".bed"
