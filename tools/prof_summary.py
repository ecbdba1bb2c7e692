#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel calls / total / average duration."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(path)))
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = name.split("(")[0][:90]
    agg[short][0] += 1
    agg[short][1] += dur
tot = sum(v[1] for v in agg.values())
print("%-92s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-92s %8d %12.1f %10.2f %6.2f" % (k, n, t, t / n, 100 * t / tot))
print("TOTAL %.1f us over %d dispatches" % (tot, len(rows)))
