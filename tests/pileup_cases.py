"""The reference's own pileup integration cases (/root/reference/tests/test_pileup.rs), as
(name, flags, input BAM, golden bedMethyl) over the data fixtures in tests/golden/modkit_fixtures.
Shared by the oracle pinning tests (CPU) and the HIP parity tests (GPU)."""
import os

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modkit_fixtures")
_CG = "CG_5mC_20230207_1700_6A_PAG66026_3c0abf27_oligo_741_adapters_modcalls_0th_sort_10_reads"
BC = "bc_anchored_10_reads.sorted.bam"
REF = os.path.join(FIX, "CGI_ladder_3.6kb_ref.fa")
BED = os.path.join(FIX, "CGI_ladder_3.6kb_ref_include_positions.bed")
_MOT = ["--motif", "CG", "0", "--motif", "CGCG", "2", "--mixed-delim", "--no-filtering", "--ref", REF,
        "--region", "oligo_741_adapters:22-62"]

# (test name in tests/test_pileup.rs : line) -> case
GOLDEN_CASES = [
    ("test_pileup_no_filt:23", ["-i", "25", "--no-filtering", "--only-tabs"], BC, "modbam.modpileup_nofilt.methyl.bed"),
    ("test_pileup_with_header:900", ["-i", "25", "--no-filtering", "--with-header"], BC, "pileup_with_header.bed"),
    ("test_pileup_with_filt:44", ["-i", "25", "-f", "1.0", "-p", "0.25", "--only-tabs", "--seed", "42",
                                  "--include-unmapped"], BC, "modbam.modpileup_filt025.methyl.bed"),
    ("test_pileup_combine:71", ["--combine-mods", "--no-filtering", "--only-tabs"], BC,
     "modbam.modpileup_combined.methyl.bed"),
    ("test_pileup_with_region:194", ["--region", "oligo_1512_adapters:0-50", "--no-filtering", "--mixed-delim"], BC,
     "modbam.modpileup_nofilt_oligo_1512_adapters_10_50.bed"),
    ("test_pileup_duplex_reads:217", ["--region", "chr17", "--no-filtering", "--mixed-delim"],
     "duplex_modbam.sorted.bam", "duplex_modbam_pileup_nofilt.bed"),
    ("test_pileup_cpg_motif_filtering:237", ["--no-filtering", "--mixed-delim", "--cpg", "--ref", REF], BC,
     "bc_anchored_10_reads_nofilt_cg_motif.bed"),
    ("test_pileup_edge_filter_regression:360", ["--no-filtering", "--mixed-delim", "--edge-filter", "50"], BC,
     "bc_anchored_10_reads_edge_filter50.bed"),
    ("test_pileup_edge_filter_asymmetric_regression:418", ["--no-filtering", "--mixed-delim", "--edge-filter", "50,0"],
     BC, "bc_anchored_10_reads_edge_filter50-0.bed"),
    ("test_pileup_with_filt_position_filter:639", ["--mixed-delim", "-i", "25", "-p", "0.25", "--include-positions", BED],
     BC, "modbam.modpileup_filt_positions_025.methyl.bed"),
    ("test_pileup_with_filter_positions_and_traditional:663",
     ["--mixed-delim", "-i", "25", "-p", "0.25", "--include-positions", BED, "--preset", "traditional", "--ref", REF],
     BC, "modbam.modpileup_filt_positions_025_traditional.methyl.bed"),
    ("test_pileup_motifs_cg0_cgcg2:738a", _MOT, _CG + ".bam", "cgcg2_cg0_test1.bed"),
    ("test_pileup_motifs_cg0_cgcg2:738b", _MOT, _CG + "-2.bam", "cgcg2_cg0_test2.bed"),
    ("test_pileup_motifs_cg0_cgcg2_combined:779a", _MOT + ["--combine-strands"], _CG + ".bam",
     "cgcg2_cg0_test1_combine_strands.bed"),
    ("test_pileup_motifs_cg0_cgcg2_combined:779b", _MOT + ["--combine-strands"], _CG + "-2.bam",
     "cgcg2_cg0_test2_combine_strands.bed"),
] + [
    ("test_pileup_cpg_motif_filtering_strand_combine:257[i=%s]" % isz,
     ["--no-filtering", "--mixed-delim", "-i", isz, "--cpg", "--combine-strands", "--ref", REF], BC,
     "bc_anchored_10_reads_nofilt_cg_motif_strand_combine.bed")
    for isz in ["10", "88", "89", "90", "91", "92", "93", "94", "10000"]
]


def fixture(name):
    return os.path.join(FIX, name)


# /root/reference/tests/test_pileup_hemi.rs: (name, flags without -r/-o, input BAM, golden)
HEMI_BAM = "duplex_modcalls_sort.bam"
_HEMI_REGION = ["--region", "chr20:22,613,835-22,640,468"]
HEMI_GOLDEN_CASES = [
    ("test_pileup_hemi_hm:13", ["--motif", "CG", "0"] + _HEMI_REGION + ["--no-filtering", "--mixed-delim"], HEMI_BAM,
     "duplex_hemi_nofilt.bed"),
    ("test_pileup_hemi_preset:42", ["--cpg"] + _HEMI_REGION + ["--mixed-delim"], HEMI_BAM, "duplex_hemi.bed"),
]


def hemi_reference_fasta(directory):
    """GRCh38_chr20.fa stand-in for the pileup-hemi cases: chr20 at its full length, N everywhere except the slice that
    tests/golden/make_hemi_reference.py rebuilt from the MD tags of the test BAM.  Written once per directory."""
    path = os.path.join(str(directory), "chr20_hemi.fa")
    if not os.path.exists(path):
        with open(fixture("chr20_hemi_slice.txt")) as f:
            name, length, start = f.readline().split()
            seq = f.readline().strip().encode()
        full = bytearray(b"N" * int(length))
        full[int(start):int(start) + len(seq)] = seq
        with open(path + ".tmp", "wb") as f:
            f.write(b">" + name.encode() + b"\n")
            for i in range(0, len(full), 100000):
                f.write(full[i:i + 100000] + b"\n")
        os.replace(path + ".tmp", path)
    return path


def update_tags_ambiguous(src_bam, dst_bam):
    """What `modkit update-tags --mode ambiguous --no-implicit-probs` (src/commands.rs:1239-1282) does to a record whose
    tags are old-style `Mm:Z:C+m,d..;` / `Ml:B:C`: the explicitly listed calls are kept with their qualities, the mode
    becomes explicit ('?'), and the tags are written under the current names `MM` / `ML` at the end of the aux block
    (remove_aux + push_aux).  Only that tag shape is handled (it is the only one in the fixture this serves,
    tests/test_pileup.rs:161-192); anything else raises."""
    import gzip
    import struct
    from bamfuzz import bgzf_write
    d = gzip.open(src_bam).read()
    o = 4
    lt, = struct.unpack_from("<i", d, o); o += 4 + lt
    nr, = struct.unpack_from("<i", d, o); o += 4
    for _ in range(nr):
        ln, = struct.unpack_from("<i", d, o); o += 4 + ln + 4
    out = bytearray(d[:o])
    width = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        rec = d[o + 4:o + 4 + bs]; o += 4 + bs
        lrn, ncig, lseq = rec[8], struct.unpack_from("<H", rec, 12)[0], struct.unpack_from("<i", rec, 16)[0]
        a = 32 + lrn + 4 * ncig + (lseq + 1) // 2 + lseq
        aux, keep, mm, ml, p = rec[a:], bytearray(), None, None, 0
        while p < len(aux):
            tag, ty, q = aux[p:p + 2], chr(aux[p + 2]), p + 3
            if ty in width:
                q += width[ty]
            elif ty in "ZH":
                q = aux.index(b"\0", q) + 1
            elif ty == "B":
                cnt, = struct.unpack_from("<i", aux, q + 1)
                q += 5 + cnt * width[chr(aux[q])]
            else:
                raise ValueError("aux type " + ty)
            if tag in (b"Mm", b"MM"):
                mm = aux[p + 3:q - 1].decode()
            elif tag in (b"Ml", b"ML"):
                assert aux[p + 2:p + 4] == b"BC"
                ml = aux[p + 8:q]
            else:
                keep += aux[p:q]
            p = q
        assert mm is not None and ml is not None
        new_mm = ""
        for part in [x for x in mm.split(";") if x]:
            head, _, rest = part.partition(",")
            assert head == "C+m", head   # default mode, one code: the explicit positions stay, the mode becomes '?'
            new_mm += "C+m?" + ("," + rest if rest else "") + ";"
        keep += b"MMZ" + new_mm.encode() + b"\0" + b"MLBC" + struct.pack("<I", len(ml)) + ml
        body = rec[:a] + bytes(keep)
        out += struct.pack("<i", len(body)) + body
    bgzf_write(dst_bam, bytes(out))
    return dst_bam

# `modkit extract calls` (tests/test_extract.rs:499-560): (name, flags, bam, golden tsv)
EXTRACT_CALLS_CASES = [
    ("extract_calls_regression", ["--ref", "{ref}"], "2_reads_all_context.bam", "test_read_calls_estimate_thresh.tsv"),
    ("extract_supplementary_secondary_calls", ["--allow-non-primary"], "supplementary_and_secondary_read.bam", "test_supplementary_calls.tsv"),
]


def convert_mod_code(src_bam, dst_bam, from_code, to_code):
    """What `modkit adjust-mods --convert <from> <to>` writes for a record whose tags are `C+h?,d;C+m?,d;` (src/adjust.rs ->
    format_mm_ml_tag, src/mod_bam.rs:1299-1387): one tag per code in ModCodeRepr order (letters by code point, then ChEBI numbers), each
    with the same delta list, the ML blocks in tag order; probabilities survive the f32 round trip ((q + 0.5) / 256 -> q).  Only that
    tag shape is handled (the fixture of tests/test_pileup.rs:373-444); anything else raises."""
    import gzip
    import struct
    from bamfuzz import bgzf_write
    d = gzip.open(src_bam).read()
    o = 4
    lt, = struct.unpack_from("<i", d, o); o += 4 + lt
    nr, = struct.unpack_from("<i", d, o); o += 4
    for _ in range(nr):
        ln, = struct.unpack_from("<i", d, o); o += 4 + ln + 4
    out = bytearray(d[:o])
    width = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    order = lambda code: (1, int(code)) if code.isdigit() else (0, ord(code))   # derive(Ord) on enum ModCodeRepr { Code(char), ChEbi(u32) }
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        rec = d[o + 4:o + 4 + bs]; o += 4 + bs
        lrn, ncig, lseq = rec[8], struct.unpack_from("<H", rec, 12)[0], struct.unpack_from("<i", rec, 16)[0]
        a = 32 + lrn + 4 * ncig + (lseq + 1) // 2 + lseq
        aux, keep, mm, ml, p = rec[a:], bytearray(), None, None, 0
        while p < len(aux):
            tag, ty, q = aux[p:p + 2], chr(aux[p + 2]), p + 3
            if ty in width:
                q += width[ty]
            elif ty in "ZH":
                q = aux.index(b"\0", q) + 1
            elif ty == "B":
                cnt, = struct.unpack_from("<i", aux, q + 1)
                q += 5 + cnt * width[chr(aux[q])]
            else:
                raise ValueError("aux type " + ty)
            if tag == b"MM":
                mm = aux[p + 3:q - 1].decode()
            elif tag == b"ML":
                assert aux[p + 2:p + 4] == b"BC"
                ml = aux[p + 8:q]
            else:
                keep += aux[p:q]
            p = q
        if mm is None:   # a record without tags stays as it is
            out += struct.pack("<i", len(rec)) + rec
            continue
        tags, at = [], 0
        for part in [x for x in mm.split(";") if x]:
            head, _, rest = part.partition(",")
            assert len(head) == 4 and head[:2] == "C+" and head[3] == "?", head
            n = len(rest.split(",")) if rest else 0
            code = to_code if head[2] == from_code else head[2]
            tags.append((code, rest, ml[at:at + n])); at += n
        assert at == len(ml) and len({t[0] for t in tags}) == len(tags)
        tags.sort(key=lambda t: order(t[0]))
        new_mm = "".join("C+%s?%s;" % (c, "," + r if r else "") for c, r, _ in tags)
        new_ml = b"".join(bytes(m) for _, _, m in tags)
        keep += b"MMZ" + new_mm.encode() + b"\0" + b"MLBC" + struct.pack("<I", len(new_ml)) + new_ml
        body = rec[:a] + bytes(keep)
        out += struct.pack("<i", len(body)) + body
    bgzf_write(dst_bam, bytes(out))
    return dst_bam


def collapse_ignore(src_bam, dst_bam, ignore="h"):
    """What `modkit adjust-mods --ignore <code>` writes for a record whose tags are `C+h?,d;C+m?,d;` over ONE delta list (src/adjust.rs ->
    BaseModProbs::into_collapsed, CollapseMethod::ReDistribute, src/mod_bam.rs:559-597; format_mm_ml_tag, 1299-1387; prob_to_qual,
    798-806): per call the ignored code's probability is split evenly between the other codes and the canonical base
    (p' = p + p_ignored / (others + 1), in f32), the ignored tag disappears, and the survivors are written back with ML = floor(256 p')
    (255 for p' == 1).  Only that tag shape is handled (the fixture of tests/test_pileup.rs:91-141); anything else raises."""
    import gzip
    import struct
    import numpy as np
    from bamfuzz import bgzf_write
    d = gzip.open(src_bam).read()
    o = 4
    lt, = struct.unpack_from("<i", d, o); o += 4 + lt
    nr, = struct.unpack_from("<i", d, o); o += 4
    for _ in range(nr):
        ln, = struct.unpack_from("<i", d, o); o += 4 + ln + 4
    out = bytearray(d[:o])
    width = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    f32 = np.float32
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        rec = d[o + 4:o + 4 + bs]; o += 4 + bs
        lrn, ncig, lseq = rec[8], struct.unpack_from("<H", rec, 12)[0], struct.unpack_from("<i", rec, 16)[0]
        a = 32 + lrn + 4 * ncig + (lseq + 1) // 2 + lseq
        aux, keep, mm, ml, p = rec[a:], bytearray(), None, None, 0
        while p < len(aux):
            tag, ty, q = aux[p:p + 2], chr(aux[p + 2]), p + 3
            if ty in width:
                q += width[ty]
            elif ty in "ZH":
                q = aux.index(b"\0", q) + 1
            elif ty == "B":
                cnt, = struct.unpack_from("<i", aux, q + 1)
                q += 5 + cnt * width[chr(aux[q])]
            else:
                raise ValueError("aux type " + ty)
            if tag == b"MM":
                mm = aux[p + 3:q - 1].decode()
            elif tag == b"ML":
                assert aux[p + 2:p + 4] == b"BC"
                ml = aux[p + 8:q]
            else:
                keep += aux[p:q]
            p = q
        if mm is None:
            out += struct.pack("<i", len(rec)) + rec
            continue
        tags, at = [], 0
        for part in [x for x in mm.split(";") if x]:
            head, _, rest = part.partition(",")
            assert len(head) == 4 and head[:2] == "C+" and head[3] == "?", head
            n = len(rest.split(",")) if rest else 0
            tags.append((head[2], rest, ml[at:at + n])); at += n
        assert at == len(ml) and len({t[1] for t in tags}) == 1, "one delta list expected"
        gone = [t for t in tags if t[0] == ignore]
        stay = sorted((t for t in tags if t[0] != ignore), key=lambda t: ord(t[0]))
        assert len(gone) == 1 and stay
        share = f32(len(stay) + 1)
        new_ml = bytearray()
        for code, rest, quals in stay:
            for k, qv in enumerate(quals):
                prob = (f32(qv) + f32(0.5)) / f32(256)
                pig = (f32(gone[0][2][k]) + f32(0.5)) / f32(256)
                newp = f32(prob + f32(pig / share))
                new_ml.append(255 if newp == f32(1.0) else int(np.floor(f32(newp * f32(256)))))
        new_mm = "".join("C+%s?%s;" % (c, "," + r if r else "") for c, r, _ in stay)
        keep += b"MMZ" + new_mm.encode() + b"\0" + b"MLBC" + struct.pack("<I", len(new_ml)) + bytes(new_ml)
        body = rec[:a] + bytes(keep)
        out += struct.pack("<i", len(body)) + body
    bgzf_write(dst_bam, bytes(out))
    return dst_bam


def chebi_case_expected_rows(golden_path, to_code):
    """tests/test_pileup.rs:373-444: the no-filter golden with `h` renamed, both sides sorted by (chrom, start, code) as the test does
    (ModCodeRepr order: letters before ChEBI numbers)."""
    key = lambda f: (f[0], int(f[1]), (1, int(f[3])) if f[3].isdigit() else (0, ord(f[3])))
    rows = [l.split("\t") for l in open(golden_path).read().splitlines()]
    for f in rows:
        if f[3] == "h":
            f[3] = to_code
    return ["\t".join(f) for f in sorted(rows, key=key)], key
