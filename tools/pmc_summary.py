#!/usr/bin/env python
"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv:  python tools/pmc_summary.py <dir> [substr ...]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    if subs and not any(s in k for s in subs):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:32s} {x / cnt[k][c]:16.0f}")
