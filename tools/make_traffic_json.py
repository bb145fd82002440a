#!/usr/bin/env python
"""profiles/hbm_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/gpu_check.sh.
Per MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes a
coalesced stream fetches, so it is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import json
import sys


def read(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        k, n, v = line.rsplit(",", 2)
        out[k] = float(v)
    return out


fetch, write, tag = read(sys.argv[1]), read(sys.argv[2]), sys.argv[3]
res = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 1`, %s; bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024" % tag,
       "_fetch_kib_per_launch": {k: v for k, v in fetch.items() if k.startswith("mkp_")},
       "_write_kib_per_launch": {k: v for k, v in write.items() if k.startswith("mkp_")}}
for k in fetch:
    if k.startswith("mkp_"):
        res[k] = int((2.0 * fetch[k] + write.get(k, 0.0)) * 1024)
json.dump(res, open("profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
