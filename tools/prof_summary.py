#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel calls / total / average duration.

  python tools/prof_summary.py <kernel_trace.csv> [--after KERNEL_SUBSTR N]

--after k_soft_ce 6 keeps only dispatches that START after the N-th launch of a kernel whose name contains
KERNEL_SUBSTR has ended (bench.py launches k_soft_ce twice per step, so N = 2*warmup drops the warm-up
steps and MIOpen's one-off solver search that runs inside the first of them)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
after, nth = None, 0
if "--after" in sys.argv:
    i = sys.argv.index("--after")
    after, nth = sys.argv[i + 1], int(sys.argv[i + 2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = 0
if after:
    seen = 0
    for r in rows:
        if after in r["Kernel_Name"]:
            seen += 1
            if seen == nth:
                t0 = int(r["End_Timestamp"])
                break
agg = defaultdict(lambda: [0, 0.0])
kept = 0
first = last = None
for r in rows:
    if int(r["Start_Timestamp"]) < t0:
        continue
    kept += 1
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    first = s if first is None else first
    last = e if last is None or e > last else last
    short = r["Kernel_Name"].split("(")[0][:90]
    agg[short][0] += 1
    agg[short][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("%-92s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%-92s %8d %12.1f %10.2f %6.2f" % (k, n, t, t / n, 100 * t / tot))
print("TOTAL kernel time %.1f us over %d dispatches; wall span %.1f us (kernels on two streams overlap)" % (tot, kept, (last - first) / 1e3 if kept else 0))
