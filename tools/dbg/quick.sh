#!/bin/bash
# quick GPU validation: parity suite on 8 workers + bench line (no cpu baseline)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -x -n 8 2>&1 | tail -n 4
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('ms_per_step', round(r['ms_per_step'], 4), r['config']['kernel_ms'], 'value', r['value'])
    else: print(l.rstrip())
"
