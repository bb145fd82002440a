#!/usr/bin/env python3
"""Rebuild the slice of GRCh38 chr20 that modkit's pileup-hemi tests need (tests/test_pileup_hemi.rs uses
tests/resources/GRCh38_chr20.fa, which is not shipped with the reference checkout) from the test BAM itself:
every record of duplex_modcalls_sort.bam carries an MD tag, so SEQ + CIGAR + MD give the exact reference base at
every position a read covers; positions no read covers have no pileup column and are written as N.

Writes tests/golden/modkit_fixtures/chr20_hemi_slice.txt: "<contig> <contig length> <slice start>" then the bases.
tests build the FASTA (N-padded to the contig length) from it at run time.

usage: make_hemi_reference.py <duplex_modcalls_sort.bam> <out.txt> [start end]"""
import gzip, re, struct, sys

def records(path):
    data = gzip.open(path, "rb").read()
    l_text, = struct.unpack_from("<i", data, 4); o = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, o); o += 4
    refs = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, o); o += 4
        name = data[o:o + l - 1].decode(); o += l
        ln, = struct.unpack_from("<i", data, o); o += 4
        refs.append((name, ln))
    while o < len(data):
        bs, = struct.unpack_from("<i", data, o); o += 4
        tid, pos, l_rn, _mq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", data, o)
        p = o + 32 + l_rn
        cig = [(c & 15, c >> 4) for c in struct.unpack_from("<%dI" % n_cig, data, p)]; p += 4 * n_cig
        sb = data[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2 + l_seq
        seq = "".join("=ACMGRSVTWYHKDBN"[(sb[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        md = None
        while p < o + bs:
            tag, t = data[p:p + 2], chr(data[p + 2]); p += 3
            if t in "AcC": p += 1
            elif t in "sS": p += 2
            elif t in "iIf": p += 4
            elif t in "ZH":
                e = data.index(b"\0", p)
                if tag == b"MD": md = data[p:e].decode()
                p = e + 1
            elif t == "B":
                st, n = chr(data[p]), struct.unpack_from("<i", data, p + 1)[0]
                p += 5 + n * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[st]
        yield refs, tid, pos, flag, cig, seq, md
        o += bs

def main():
    bam, out = sys.argv[1], sys.argv[2]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (22_613_000, 22_641_500)
    ref = bytearray(b"N" * (hi - lo))
    contig = None
    for refs, tid, pos, flag, cig, seq, md in records(bam):
        if flag & 4 or md is None: continue
        contig = refs[tid]
        toks = re.findall(r"(\d+)|(\^[A-Za-z]+)|([A-Za-z])", md)
        mdq = []   # ['m', n] matches, ['x', base] mismatch, ['d', bases] deletion
        for n, d, x in toks:
            if n: mdq.append(["m", int(n)])
            elif d: mdq.append(["d", d[1:]])
            else: mdq.append(["x", x])
        mi = 0; q = 0; r = pos
        def put(p, b):
            if lo <= p < hi:
                b = b.upper()
                assert ref[p - lo] in (ord("N"), ord(b)), (p, chr(ref[p - lo]), b)
                ref[p - lo] = ord(b)
        for op, ln in cig:
            if op in (0, 7, 8):
                k = 0
                while k < ln:
                    while mdq[mi][0] == "m" and mdq[mi][1] == 0: mi += 1
                    kind, v = mdq[mi]
                    if kind == "m":
                        n = min(v, ln - k)
                        for j in range(n): put(r + j, seq[q + j])
                        mdq[mi][1] -= n; k += n; q += n; r += n
                    elif kind == "x":
                        put(r, v); mi += 1; k += 1; q += 1; r += 1
                    else: raise AssertionError("deletion inside a match op")
            elif op in (1, 4): q += ln
            elif op == 2:
                while mdq[mi][0] == "m" and mdq[mi][1] == 0: mi += 1
                assert mdq[mi][0] == "d" and len(mdq[mi][1]) == ln, (mdq[mi], ln)
                for j in range(ln): put(r + j, mdq[mi][1][j])
                mi += 1; r += ln
            elif op == 3: r += ln
    with open(out, "w") as f:
        f.write("%s %d %d\n" % (contig[0], contig[1], lo))
        f.write(ref.decode() + "\n")
    print("covered %d of %d positions" % (sum(1 for b in ref if b != ord("N")), len(ref)))

if __name__ == "__main__":
    main()
