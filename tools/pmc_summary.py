#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_summary.py <dir> [kernel-substring ...]"""
import csv, glob, sys, collections
root = sys.argv[1]
filters = sys.argv[2:] or [""]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if any(s in name for s in filters):
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    print(name[:100])
    for c, v in sorted(cs.items()):
        print("   %-28s avg %14.1f  over %d dispatches" % (c, sum(v) / len(v), len(v)))
