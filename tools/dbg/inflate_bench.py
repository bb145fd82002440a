#!/usr/bin/env python
"""Device BGZF inflate throughput on the bench BAM (mkp_bgzf_inflate; every block CRC-checked on the host inside the call).
Usage: python tools/dbg/inflate_bench.py [bam]   (default: generates the C3 bench BAM under /tmp)"""
import ctypes, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import modkit_amd
if len(sys.argv) > 1:
    bam = sys.argv[1]
else:
    bam = "/tmp/inflate_c3.bam"
    if not os.path.exists(bam):
        subprocess.check_call([os.path.join(ROOT, "tools", "gen_modbam"), "--out", "/tmp/inflate_c3", "--contig", "chr20:64444167", "--reads", "193000", "--seed", "20", "--style", "hm",
                               "--cpg-depleted", "--mean-len", "8353", "--threads", str(os.cpu_count() or 8)], stdout=subprocess.DEVNULL)
data = open(bam, "rb").read()
if os.environ.get("INFLATE_BLOCKS"):   # the first N blocks only (launch-size sweeps)
    o, k = 0, 0
    while o < len(data) and k < int(os.environ["INFLATE_BLOCKS"]):
        o += int.from_bytes(data[o + 16:o + 18], "little") + 1; k += 1
    data = data[:o]
ctx = modkit_amd.Context()
res = []
for rep in range(3):
    out, n, ms = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_double()
    t0 = time.time()
    ctx._check(ctx.L.mkp_bgzf_inflate(ctx.h, data, len(data), ctypes.byref(out), ctypes.byref(n), ctypes.byref(ms)))
    res.append({"kernel_ms": ms.value, "call_s": time.time() - t0})
print(json.dumps({"bam_bytes": len(data), "inflated_bytes": n.value, "runs": res,
                  "kernel_GBps_inflated": n.value / (min(r["kernel_ms"] for r in res) * 1e-3) / 1e9,
                  "kernel_GBps_compressed": len(data) / (min(r["kernel_ms"] for r in res) * 1e-3) / 1e9}))
ctx.close()
