"""The host DEFLATE decoder of the BGZF ingest (modkit_amd/csrc/mkp_inflate_host.hpp) against zlib: every block type (stored, fixed,
dynamic), every compression level and strategy, short-distance matches, long literal runs, multi-block streams, the BGZF size limit,
and malformed input (declined, never written out of bounds)."""
import ctypes
import os
import random
import zlib

import numpy as np
import pytest

import modkit_amd


def host_inflate(payload: bytes, dlen: int):
    L = modkit_amd.lib()
    L.mkp_internal_host_inflate.restype = ctypes.c_int
    L.mkp_internal_host_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    guard = 64
    dst = (ctypes.c_uint8 * (dlen + guard))(*([0xA5] * (dlen + guard)))
    ok = L.mkp_internal_host_inflate(payload, len(payload), dst, dlen)
    raw = bytes(dst)
    assert raw[dlen:] == b"\xa5" * guard, "wrote past the output"
    return ok, raw[:dlen]


def raw_deflate(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, chunks=1):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    out = b""
    step = max(1, len(data) // chunks)
    for i in range(0, len(data), step):
        out += c.compress(data[i:i + step])
        if chunks > 1:
            out += c.flush(zlib.Z_FULL_FLUSH)   # ends a block (adds an empty stored block): multi-block streams
    return out + c.flush()


def corpora():
    rng = random.Random(7)
    nprng = np.random.default_rng(7)
    yield "empty", b""
    yield "one", b"A"
    yield "run", b"\x00" * 65280
    yield "pair-run", b"ab" * 30000
    yield "period7", b"abcdefg" * 9000
    yield "random", nprng.integers(0, 256, 65280, dtype=np.uint8).tobytes()
    yield "nibbles", nprng.integers(0, 16, 60000, dtype=np.uint8).tobytes()
    yield "skewed", bytes(nprng.choice(256, 65000, p=np.r_[np.full(4, 0.2), np.full(252, 0.2 / 252)]).astype(np.uint8))
    yield "text", (b"chr20\t%d\t%d\tm\t30\t+\n" * 1)[:0] + b"".join(b"chr20\t%d\t%d\tm\t%d\t+\n" % (i, i + 1, rng.randrange(100)) for i in range(2500))
    # BAM-like: names, CIGARs, packed bases, qualities, MM/ML tags
    rec = b"".join(bytes([rng.randrange(256) for _ in range(32)]) + b"read%06d\x00" % i + bytes(nprng.integers(0, 256, 300, dtype=np.uint8)) + bytes(nprng.integers(20, 45, 500, dtype=np.uint8))
                   + b"MMZC+m?," + b",".join(b"%d" % rng.randrange(30) for _ in range(60)) + b";\x00" for i in range(60))
    yield "bam-like", rec[:65280]
    for n in (1, 2, 3, 7, 8, 9, 257, 258, 259, 273, 274, 275, 1000):
        yield "short%d" % n, bytes(nprng.integers(0, 4, n, dtype=np.uint8))


@pytest.mark.parametrize("level", [0, 1, 3, 6, 9])
def test_matches_zlib_all_levels(level):
    for name, data in corpora():
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            z = raw_deflate(data, level, strategy)
            ok, got = host_inflate(z, len(data))
            assert ok == 1, (name, level, strategy)
            assert got == data, (name, level, strategy)


def test_multi_block_streams():
    for name, data in corpora():
        if len(data) < 64:
            continue
        for chunks in (2, 5, 17):
            z = raw_deflate(data, 6, zlib.Z_DEFAULT_STRATEGY, chunks)
            ok, got = host_inflate(z, len(data))
            assert ok == 1 and got == data, (name, chunks)


def test_declines_malformed():
    data = b"".join(b"%d," % (i * 7919 % 1000) for i in range(8000))
    z = raw_deflate(data)
    assert host_inflate(z, len(data))[0] == 1
    assert host_inflate(z, len(data) - 1)[0] == 0          # output size mismatch
    assert host_inflate(z, len(data) + 1)[0] == 0
    assert host_inflate(z[:len(z) // 2], len(data))[0] == 0   # truncated
    assert host_inflate(b"", 10)[0] == 0
    assert host_inflate(b"\x07", 0)[0] == 0                 # reserved block type 3
    rng = random.Random(3)
    accepted = 0
    for trial in range(600):   # random corruption: the decoder accepts exactly what zlib accepts (as a stream of this size), with the same bytes
        b = bytearray(z)
        for _ in range(rng.randrange(1, 4)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        ok, got = host_inflate(bytes(b), len(data))
        try:
            want = zlib.decompress(bytes(b), -15)
        except zlib.error:
            want = None
        if want is not None and len(want) == len(data):
            assert ok == 1 and got == want
            accepted += 1
        else:
            assert ok == 0
    assert 0 < accepted < 600


def test_random_garbage_is_safe():
    rng = np.random.default_rng(11)
    for trial in range(400):
        n = int(rng.integers(1, 400))
        host_inflate(rng.integers(0, 256, n, dtype=np.uint8).tobytes(), int(rng.integers(0, 70000)))


def test_every_block_of_the_golden_bams():
    """The ingest itself uses the decoder (zlib only when it declines): every BGZF block of the reference's test BAMs through both."""
    import glob
    import struct
    n = 0
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "**", "*.bam"), recursive=True)):
        raw = open(path, "rb").read()
        o = 0
        while o + 18 <= len(raw):
            xlen = struct.unpack_from("<H", raw, o + 10)[0]
            bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
            payload = raw[o + 12 + xlen:o + bsize - 8]
            isize = struct.unpack_from("<I", raw, o + bsize - 4)[0]
            want = zlib.decompress(payload, -15)
            assert len(want) == isize
            ok, got = host_inflate(payload, isize)
            assert ok == 1 and got == want, (path, o)
            o += bsize
            n += 1
    assert n > 50


def test_crc32_equals_zlib():
    """mkp_crc32.hpp (PCLMULQDQ folding where the host has it) against zlib.crc32: every length around the 16 / 64-byte step sizes, random
    offsets (unaligned starts), block-sized inputs."""
    L = modkit_amd.lib()
    L.mkp_internal_crc32.restype = ctypes.c_uint32
    L.mkp_internal_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    rng = np.random.default_rng(11)
    buf = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    lens = list(range(0, 300)) + [1023, 1024, 1025, 65279, 65280, 65281, 65536] + [int(x) for x in rng.integers(300, 66000, 200)]
    for n in lens:
        off = int(rng.integers(0, 64))
        piece = buf[off:off + n]
        assert L.mkp_internal_crc32(piece, len(piece)) == (zlib.crc32(piece) & 0xffffffff), (n, off)
