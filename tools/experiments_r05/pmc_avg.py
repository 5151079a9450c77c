#!/usr/bin/env python3
"""Average of PMC counters per kernel-name substring from a rocprofv3 --pmc run directory (counter_collection.csv).
  python tools/experiments_r05/pmc_avg.py DIR SUBSTR [SUBSTR ...]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        for s in subs:
            if s in k:
                a = acc[(s, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
for (s, c), (tot, n) in sorted(acc.items()):
    print("%-24s %-16s avg %.1f over %d dispatches" % (s, c, tot / n, n))
