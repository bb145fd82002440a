"""The --bedgraph writer on the CPU (mkp_internal_bedgraph_write: rows in, files out): the rows of the reference's golden bedMethyl files
go through it and must come out as the projection BedGraphWriter makes of them (src/writers.rs:264-381) — file per (key, strand, code,
motif), `chrom pos pos+1 fraction coverage`, the fraction an f32 through Rust's `{}`.  The device runs are tests/test_gpu_bedgraph.py."""
import ctypes
import glob
import os

import numpy as np
import pytest

import modkit_amd
from pileup_cases import FIX

STRAND = {"+": "positive", "-": "negative", ".": "combined"}


def rust_f32(n_mod, valid):
    return np.format_float_positional(np.float32(n_mod) / np.float32(valid), unique=True, trim="-")


def rows_of(lines, labels, keys=None):
    """bedMethyl lines -> a mkp_rows (and what keeps its arrays alive)."""
    n = len(lines)
    cols = {k: np.zeros(n, dtype=np.uint32) for k in ("pos", "code", "valid", "mod", "key")}
    strand = np.zeros(n, dtype=np.uint8)
    motif = np.full(n, -1, dtype=np.int32)
    for i, line in enumerate(lines):
        f = line.split("\t")
        name = f[3].split(",", 1)
        cols["pos"][i] = int(f[1])
        cols["code"][i] = (0x80000000 | int(name[0])) if name[0].isdigit() else ord(name[0])
        if len(name) > 1:
            motif[i] = labels.index(name[1])
        strand[i] = ord(f[5])
        cols["valid"][i], cols["mod"][i] = int(f[9]), int(f[11])
        if keys:
            cols["key"][i] = i % len(keys)
    r = modkit_amd.Rows()
    r.n_rows = n
    u32 = ctypes.POINTER(ctypes.c_uint32)
    r.pos = cols["pos"].ctypes.data_as(u32)
    r.code_repr = cols["code"].ctypes.data_as(u32)
    r.n_valid = cols["valid"].ctypes.data_as(u32)
    r.n_mod = cols["mod"].ctypes.data_as(u32)
    r.strand = strand.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    r.motif_idx = motif.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    keep = [cols, strand, motif]
    if keys:
        names = (ctypes.c_char_p * len(keys))(*[k.encode() for k in keys])
        r.partition_key = cols["key"].ctypes.data_as(u32)
        r.n_partition_keys = len(keys)
        r.partition_key_names = names
        keep.append(names)
    return r, keep


def write(tmp, chrom_lines, labels, prefix=None, keys=None):
    L = modkit_amd.lib()
    L.mkp_internal_bedgraph_write.restype = ctypes.c_int
    L.mkp_internal_bedgraph_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p]
    lab = (ctypes.c_char_p * max(len(labels), 1))(*[x.encode() for x in labels])
    os.makedirs(tmp, exist_ok=True)
    for chrom, lines in chrom_lines:   # (the driver calls the writer once per shard; files are appended to within a run only, so one contig per call here)
        r, keep = rows_of(lines, labels, keys)
        assert L.mkp_internal_bedgraph_write(tmp.encode(), prefix.encode() if prefix else None, 1 if keys else 0, lab, len(labels), chrom.encode(), ctypes.byref(r)) == 0
    return {fn: open(os.path.join(tmp, fn)).read() for fn in sorted(os.listdir(tmp))}


@pytest.mark.parametrize("bed", sorted(os.path.basename(p) for p in glob.glob(os.path.join(FIX, "*.bed")) if os.path.getsize(p) > 0))
def test_golden_bedmethyl_rows_through_the_writer(tmp_path, bed):
    lines = [l for l in open(os.path.join(FIX, bed)).read().splitlines() if l and not l.startswith("#") and len(l.replace(" ", "\t").split("\t")) >= 18]
    lines = [l.replace(" ", "\t") for l in lines]
    if not lines:
        pytest.skip("not a bedMethyl file")
    chroms = sorted({l.split("\t")[0] for l in lines})
    if len(chroms) != 1:
        pytest.skip("several contigs: one call per shard in the driver, one file set per run")
    labels = sorted({l.split("\t")[3].split(",", 1)[1] for l in lines if "," in l.split("\t")[3]})
    got = write(str(tmp_path / "bg"), [(chroms[0], lines)], labels)
    want = {}
    for l in lines:
        f = l.split("\t")
        name = f[3].split(",", 1)
        fn = name[0] + ("_" + name[1].replace(",", "") if len(name) > 1 else "") + "_" + STRAND[f[5]] + ".bedgraph"
        want[fn] = want.get(fn, "") + "%s\t%s\t%d\t%s\t%s\n" % (f[0], f[1], int(f[1]) + 1, rust_f32(int(f[11]), int(f[9])), f[9])
    assert got == want and len(got) >= 1


def test_prefix_keys_chebi_and_known_fractions(tmp_path):
    lines = ["c\t10\t11\tm\t3\t+\t10\t11\t255,0,0\t3\t33.33\t1\t2\t0\t0\t0\t0\t0",
             "c\t11\t12\t76792\t65535\t-\t11\t12\t255,0,0\t65535\t0.00\t1\t65534\t0\t0\t0\t0\t0",
             "c\t12\t13\tm,CG,0\t2\t.\t12\t13\t255,0,0\t2\t100.00\t2\t0\t0\t0\t0\t0\t0"]
    got = write(str(tmp_path / "bg"), [("c", lines)], ["CG,0"], prefix="pre", keys=["ungrouped", "A_1"])
    assert got == {"pre_ungrouped_m_positive.bedgraph": "c\t10\t11\t0.33333334\t3\n",
                   "pre_A_1_76792_negative.bedgraph": "c\t11\t12\t0.000015259022\t65535\n",
                   "pre_ungrouped_m_CG0_combined.bedgraph": "c\t12\t13\t1\t2\n"}
