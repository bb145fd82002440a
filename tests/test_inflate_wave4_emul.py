"""mkp_inflate_wave4 (one wave per BGZF block: speculative token decode, a scalar walk that marks the chain, a parallel output step with in-pass
pointer jumping; 4 KiB ring + far reads) checked on the CPU: tests/inflate_wave4_emul.cpp restates the kernel's control flow over 64 emulated
lanes around the per-lane functions the kernel itself compiles (mkp_inflate_tok.hpp) and compares every block with zlib — output and
acceptance.  The GPU run of the same corpus is tests/test_gpu_inflate.py."""
import os
import random
import struct
import subprocess
import zlib

import numpy as np
import pytest

from test_host_deflate import corpora, raw_deflate

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("w4") / "inflate_wave4_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "inflate_wave4_emul.cpp"), "-lz"], check=True)
    return exe


def deflate_corpus(seed=5, corrupt=400, full_block=False):
    """(payload, expected size) records: every corpus of test_host_deflate at every level and strategy, multi-block streams, long stored
    blocks (the input window's seek path), and corrupted copies."""
    recs = []
    for level in (0, 1, 3, 6, 9):
        for name, data in corpora():
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                recs.append((raw_deflate(data, level, strategy), len(data)))
    for name, data in corpora():
        if len(data) >= 64:
            for chunks in (2, 5, 17):
                recs.append((raw_deflate(data, 6, zlib.Z_DEFAULT_STRATEGY, chunks), len(data)))
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    # stored block in the middle of compressed ones; matches that reach back across it
    a = b"".join(b"%d\t%d\n" % (i, i * i) for i in range(3000))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = c.compress(a) + c.flush(zlib.Z_FULL_FLUSH)
    c0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    mid = nprng.integers(0, 256, 20000, dtype=np.uint8).tobytes()
    recs.append((raw_deflate(a + mid + a, 6), len(a) * 2 + len(mid)))
    recs.append((raw_deflate(mid * 3, 0), len(mid) * 3))
    if full_block:   # (CPU emulation only) the largest block BGZF allows, 65 536 bytes out: records far apart that repeat, so that matches reach back past an 8 KiB ring
        rec = nprng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
        big = b"".join(rec[:2000 + 37 * k] + b"%06d" % k for k in range(40))
        big = (big * 3)[:65536]
        recs.append((raw_deflate(big, 9), len(big)))
        recs.append((raw_deflate(big, 1, zlib.Z_FIXED), len(big)))
    base = [r for r in recs if len(r[0]) > 40]
    for trial in range(corrupt):
        z, n = base[rng.randrange(len(base))]
        b = bytearray(z)
        kind = rng.randrange(4)
        if kind == 0:
            b = b[:rng.randrange(1, len(b))]
        else:
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(min(len(b), 400 if kind == 1 else len(b)))] ^= 1 << rng.randrange(8)
        recs.append((bytes(b), n if rng.random() < 0.8 else max(0, n + rng.randrange(-3, 4))))
    for trial in range(100):
        recs.append((nprng.integers(0, 256, int(nprng.integers(1, 400)), dtype=np.uint8).tobytes(), int(nprng.integers(0, 70000))))
    recs.append((b"", 10))
    recs.append((b"\x07", 0))
    # a dynamic block that holds nothing but its end-of-block symbol: ONE literal/length code, one bit long — an incomplete code zlib accepts
    # (inflate_table: max == 1); with a distance code of length 0 and of length 1.  And the same block asked to decode its unused code.
    recs.append((bytes.fromhex("05e081080000000020f85b1f"), 0))
    recs.append((bytes.fromhex("05e081080000000020f85b3f"), 0))
    recs.append((bytes.fromhex("05e081080000000020f85b5f"), 0))   # (first data bit 1: the code that does not exist)
    return recs


def write_corpus(path, recs):
    with open(path, "wb") as f:
        for z, n in recs:
            f.write(struct.pack("<II", len(z), n))
            f.write(z)


MODES = pytest.mark.parametrize("wave", ["4096", "2048", "8192", "32768"], ids=["ring4k_the_kernel", "ring2k", "ring8k", "ring32k_no_far_reads"])   # ring sizes of the kernel body


def run_emul(emul, args, wave):
    return subprocess.run([emul] + args, capture_output=True, text=True, env=dict(os.environ, RING=wave))


@MODES
def test_every_block_of_the_golden_bams(emul, wave):
    import glob
    bams = sorted(glob.glob(os.path.join(HERE, "golden", "**", "*.bam"), recursive=True))
    p = run_emul(emul, ["bgzf"] + bams, wave)
    assert p.returncode == 0 and p.stdout.startswith("ok "), p.stderr[-500:]
    blocks, nbytes, acc, rej = map(int, p.stdout.split()[1:5])
    assert blocks > 100 and rej == 0


@MODES
def test_deflate_corpus_and_corruptions(emul, tmp_path, wave):
    recs = deflate_corpus(full_block=True)
    path = str(tmp_path / "corpus.bin")
    write_corpus(path, recs)
    p = run_emul(emul, ["corpus", path], wave)
    assert p.returncode == 0 and p.stdout.startswith("ok "), p.stderr[-800:]
    blocks, nbytes, acc, rej = map(int, p.stdout.split()[1:5])
    assert blocks == len(recs) and acc > 700 and rej > 200, p.stdout


@MODES
def test_fuzzed_bams(emul, tmp_path, wave):
    from bamfuzz import Fuzz
    paths = []
    for seed, prof in ((3, "hm_split"), (4, "duplex_hm"), (5, "a_only")):
        try:
            bam, _, _ = Fuzz(seed, contigs=(("c", 300000),), n_reads=1500, mean_len=3000, profile=prof).write(str(tmp_path / ("f%d" % seed)))
        except Exception:
            bam, _, _ = Fuzz(seed, contigs=(("c", 300000),), n_reads=1500, mean_len=3000, profile="hm_split").write(str(tmp_path / ("f%d" % seed)))
        paths.append(bam)
    p = run_emul(emul, ["bgzf"] + paths, wave)
    assert p.returncode == 0 and p.stdout.startswith("ok "), p.stderr[-500:]
    if wave != "32768":
        assert "far reads" in p.stderr and int(p.stderr.split("far reads ")[1].split()[0]) > 1000   # (the far path is exercised)
