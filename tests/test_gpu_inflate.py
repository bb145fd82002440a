"""Device BGZF inflate (mkp_bgzf_inflate -> mkp_inflate_wave4; SURVEY §8 f1 first stage) against Python's gzip
on the reference's BAM fixtures and on generated BAMs; corrupt input must come back as an error."""
import glob
import gzip
import os

import pytest

import modkit_amd
from bamfuzz import Fuzz
from pileup_cases import FIX

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = modkit_amd.Context()
    yield c
    c.close()


def test_fixture_bams_inflate_like_gzip(ctx):
    bams = sorted(glob.glob(os.path.join(FIX, "*.bam")))
    assert len(bams) >= 8
    for b in bams:
        data = open(b, "rb").read()
        got, ms = ctx.bgzf_inflate(data)
        assert got == gzip.decompress(data), b
        assert ms >= 0


def test_generated_bam_and_empty_input(ctx, tmp_path):
    bam, _, _ = Fuzz(77, contigs=(("c", 400000),), n_reads=4000, mean_len=3000, profile="hm_split", weird_rate=0.0).write(str(tmp_path / "big"))
    data = open(bam, "rb").read()
    got, ms = ctx.bgzf_inflate(data)
    want = gzip.decompress(data)
    assert got == want and len(want) > 20_000_000
    assert ctx.bgzf_inflate(b"")[0] == b""


def test_corrupt_blocks_are_refused(ctx):
    data = bytearray(open(os.path.join(FIX, "bc_anchored_10_reads.sorted.bam"), "rb").read())
    for at in (40, 200, 1500, 3000):   # payload bytes of the first block
        bad = bytearray(data)
        bad[at] ^= 0x5A
        with pytest.raises(modkit_amd.MkpError):
            ctx.bgzf_inflate(bytes(bad))
    with pytest.raises(modkit_amd.MkpError):
        ctx.bgzf_inflate(bytes(data[:len(data) // 2]))   # cut inside a block
    with pytest.raises(modkit_amd.MkpError):
        ctx.bgzf_inflate(b"not a bgzf file at all.." * 4)


def test_device_inflate_on_the_fetch_path(tmp_path):
    """`pileup --device-inflate`: the shard windows' blocks go through the device decoder (mkp_internal_device_inflate) instead of the host
    pool; the bedMethyl must not change and the --stats line must say how many bytes took the device route."""
    import re
    import subprocess
    from test_host_ingest import gen   # tools/gen_modbam: writes the BAI the indexed fetch needs
    bam, fa = gen(tmp_path, "dv", [("c", 600000)], 12000, ["--mean-len", "4000"])
    modkit_amd.build()
    cli = os.path.join(os.path.dirname(modkit_amd.LIB_PATH), "mkpileup")
    outs = []
    for extra in (["--host-ingest"], ["--device-inflate"]):
        out = str(tmp_path / ("o%d.bed" % len(outs)))
        p = subprocess.run([cli, "pileup", bam, out, "--cpg", "--ref", fa, "--stats"] + extra, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        outs.append((open(out, "rb").read(), int(re.search(r"on the device (\d+)", p.stderr).group(1)), int(re.search(r"bam_bytes_inflated=(\d+)", p.stderr).group(1))))
    assert outs[0][0] == outs[1][0] and len(outs[0][0]) > 100000
    assert outs[0][1] == 0 and outs[1][1] > 0.5 * outs[1][2] and outs[1][1] > 32 << 20   # (the threshold sampler's 2 MiB heads stay on the host)


def bgzf_block(payload: bytes, isize: int, crc: int) -> bytes:
    import struct
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(payload) + 25) + payload + struct.pack("<II", crc & 0xffffffff, isize))


KERNEL_SCRIPT = """
import glob, gzip, os, pickle, sys, zlib
sys.path.insert(0, %r)
import modkit_amd
from test_gpu_inflate import bgzf_block
c = modkit_amd.Context()
n = 0
for b in sorted(glob.glob(os.path.join(%r, '*.bam'))):
    d = open(b, 'rb').read(); got, _ = c.bgzf_inflate(d); assert got == gzip.decompress(d), b; n += 1
recs = pickle.load(open(%r, 'rb'))
good, want, bad = [], [], 0
for z, size in recs:
    if len(z) + 26 > 65536 or size > 65536:
        continue
    try:
        w = zlib.decompress(z, -15)
        if len(w) != size:
            w = None
    except zlib.error:
        w = None
    if w is not None:
        good.append(bgzf_block(z, size, zlib.crc32(w))); want.append(w)
    else:   # what zlib refuses (as a stream of this size) must be refused here, whatever the CRC field says
        try:
            c.bgzf_inflate(bgzf_block(z, size, 0))
        except modkit_amd.MkpError:
            bad += 1
        else:
            raise AssertionError('accepted a stream zlib refuses: %%d bytes -> %%d' %% (len(z), size))
got, _ = c.bgzf_inflate(b''.join(good))
assert got == b''.join(want)
for k in (0, len(good) // 2, len(good) - 1):   # and one at a time
    assert c.bgzf_inflate(good[k])[0] == want[k]
c.close(); print('ok', n, len(good), bad)
"""


def test_all_device_kernels(tmp_path):
    """mkp_inflate_wave4 and its ring-size variants (4 KiB: the product's; 8 KiB, 2 KiB: A/B builds), forced through MKP_INFLATE_KERNEL in
    fresh processes (the variable is read once) and checked against gzip on the reference's BAMs and against zlib on the DEFLATE corpus of
    tests/test_inflate_wave4_emul.py: every block type, level and strategy, multi-block streams, long stored blocks, and ~500 corrupted
    or random streams whose acceptance must be zlib's."""
    import pickle
    import subprocess
    import sys
    from test_inflate_wave4_emul import deflate_corpus
    recs = deflate_corpus()
    pk = str(tmp_path / "corpus.pkl")
    pickle.dump(recs, open(pk, "wb"))
    here = os.path.dirname(os.path.abspath(__file__))
    script = KERNEL_SCRIPT % (os.path.dirname(here), FIX, pk)
    for kernel in ("wave4", "wave4_8k", "wave4_2k"):
        p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(os.environ, MKP_INFLATE_KERNEL=kernel, PYTHONPATH=here + os.pathsep + os.environ.get("PYTHONPATH", "")))
        assert p.returncode == 0 and p.stdout.startswith("ok"), (kernel, p.stdout[-200:], p.stderr[-600:])
        n, good, bad = map(int, p.stdout.split()[1:4])
        assert n >= 8 and good > 700 and bad > 200, (kernel, p.stdout)
