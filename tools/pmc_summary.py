#!/usr/bin/env python
"""Summarise a rocprofv3 counter_collection.csv: per kernel, dispatches and mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
tot, n = defaultdict(float), defaultdict(int)
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
print("kernel,dispatches,mean_%s_per_dispatch" % counter)
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%s,%d,%.1f" % (k, n[k], tot[k] / n[k]))
