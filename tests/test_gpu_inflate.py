"""Device BGZF inflate (mkp_bgzf_inflate -> mkp_inflate_blocks, one thread per block; SURVEY §8 f1 first stage) against Python's gzip
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


def test_both_device_kernels(tmp_path):
    """mkp_bgzf_inflate picks its kernel by launch size (one wave per block below 24 576 blocks, one thread per block above); all three
    (wave, thread, thread2 = the second edition of the per-thread decoder) are forced here through MKP_INFLATE_KERNEL in fresh processes
    (the variable is read once) and checked against gzip."""
    import subprocess
    import sys
    script = (
        "import glob, gzip, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import modkit_amd\n"
        "c = modkit_amd.Context()\n"
        "n = 0\n"
        "for b in sorted(glob.glob(os.path.join(%r, '*.bam'))):\n"
        "    d = open(b, 'rb').read(); got, _ = c.bgzf_inflate(d); assert got == gzip.decompress(d), b; n += 1\n"
        "c.close(); print('ok', n)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), FIX)
    for kernel in ("wave", "thread", "thread2"):
        p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(os.environ, MKP_INFLATE_KERNEL=kernel))
        assert p.returncode == 0 and p.stdout.startswith("ok"), (kernel, p.stderr[-400:])
