#!/usr/bin/env python
"""Mutation fuzzer for the host side (BGZF/BAM ingest, aux scan, MM tokeniser, packer): corrupts a decompressed BAM (random
bytes, truncation, record core fields, bytes near the MM/ML/MN tags), re-wraps it as BGZF and runs `--plan-only` (no device).
Every run must end in success or a clean error; with the sanitizer build of tools/asan_host.sh any report is a failure.

    tools/asan_host.sh && python tools/mutate_bam.py tests/golden/modkit_fixtures/bc_anchored_10_reads.sorted.bam 150 1
    python tools/mutate_bam.py in.bam 150 1 modkit_amd/csrc/mkpileup      # plain build: crashes only
"""
import gzip, random, struct, subprocess, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from bamfuzz import bgzf_write
src = sys.argv[1]; n_iter = int(sys.argv[2]); seed = int(sys.argv[3])
exe = sys.argv[4] if len(sys.argv) > 4 else '/tmp/mkpileup_asan'
import tempfile
work = tempfile.mkdtemp(prefix='mkp_mut_')
raw = bytearray(gzip.open(src, 'rb').read())
# find header end
l_text = struct.unpack_from('<i', raw, 4)[0]; o = 8 + l_text; n_ref = struct.unpack_from('<i', raw, o)[0]; o += 4
for _ in range(n_ref):
    ln = struct.unpack_from('<i', raw, o)[0]; o += 4 + ln + 4
hdr_end = o
rng = random.Random(seed)
bad = 0; errs = 0; oks = 0
for it in range(n_iter):
    d = bytearray(raw)
    mode = rng.randrange(6)
    if mode == 0:    # flip random bytes in record area
        for _ in range(rng.randrange(1, 8)):
            p = rng.randrange(hdr_end, len(d)); d[p] = rng.randrange(256)
    elif mode == 1:  # truncate
        d = d[:rng.randrange(hdr_end, len(d))]
    elif mode == 2:  # corrupt a block_size / core field of a random record
        o = hdr_end; offs = []
        while o + 4 <= len(d):
            bs = struct.unpack_from('<i', d, o)[0]
            if bs < 32 or o + 4 + bs > len(d): break
            offs.append(o); o += 4 + bs
        if offs:
            r = rng.choice(offs); f = rng.choice([0, 4, 8, 12, 13, 16, 18, 20, 24])
            struct.pack_into('<I', d, r + f, rng.choice([0, 1, 0xffffffff, 0x7fffffff, rng.getrandbits(32), rng.getrandbits(16)]))
    elif mode >= 4:  # corrupt near a modified-base tag
        occ = []
        for tag in (b'MMZ', b'MLB', b'MNi', b'MmZ', b'MlB', b'MNC', b'MNS'):
            s = 0
            while True:
                k = d.find(tag, s)
                if k < 0: break
                occ.append(k); s = k + 1
        if occ:
            k = rng.choice(occ)
            for _ in range(rng.randrange(1, 6)):
                p = min(len(d) - 1, k + rng.randrange(0, 48)); d[p] = rng.choice([0, 0x2c, 0x3b, 0x3f, 0x2e, 0x2b, 0x2d, 0x39, 0x43, 0x6d, 0xff, rng.randrange(256)])
    else:            # zero a run
        p = rng.randrange(hdr_end, len(d)); n = rng.randrange(1, 64); d[p:p + n] = bytes(min(n, len(d) - p))
    path = os.path.join(work, 'm.bam')
    bgzf_write(path, bytes(d))
    for flags in ([], ['--plan-pack-min', '0']):
        p = subprocess.run([exe, 'pileup', path, os.path.join(work, 'plan.tsv'), '--plan-only'] + flags, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'))
        txt = p.stdout + p.stderr
        if 'AddressSanitizer' in txt or 'runtime error' in txt or p.returncode < 0:
            bad += 1; open(os.path.join(work, 'bad_%d_%d.bam' % (seed, it)), 'wb').write(open(path, 'rb').read()); print('BAD iter', it, 'mode', mode, 'rc', p.returncode, txt[-600:]); break
        if p.returncode != 0: errs += 1
        else: oks += 1
print('done: bad', bad, 'clean errors', errs, 'ok', oks)
sys.exit(1 if bad else 0)
