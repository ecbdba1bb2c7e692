#!/usr/bin/env python3
"""GPU idle analysis of a rocprofv3 kernel trace: union of busy intervals over all streams inside the window that
starts after the N-th k_soft_ce launch; reports busy/idle time and the largest idle gaps with the kernels around them.

  python tools/prof_gaps.py <kernel_trace.csv> [N=6] [M=0]      M > 0: the window ends with the (N + M)-th k_soft_ce launch's step
  (bench.py --steps 7 --warmup 3: N = 6, M = 14 keeps the seven timed steps and drops the synchronised probe steps behind them)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen = 0; t0 = 0
for r in rows:
    if "k_soft_ce" in r["Kernel_Name"]:
        seen += 1
        if seen == nth:
            t0 = int(r["End_Timestamp"]); break
mth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t1 = 1 << 62
if mth:
    seen = 0
    for r in rows:                      # the window ends where the step AFTER the last kept one starts (its first k_soft_ce is launch N + M + 1: cut at
        if "k_soft_ce" in r["Kernel_Name"]:            # the optimizer kernel in front of it)
            seen += 1
            if seen == nth + mth + 1:
                t1 = int(r["Start_Timestamp"]); break
    last_opt = max((int(r["End_Timestamp"]) for r in rows if "k_adamw" in r["Kernel_Name"] and int(r["End_Timestamp"]) < t1), default=t1)
    t1 = last_opt
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
span = iv[-1][1] - iv[0][0]
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]; gaps = []
last_name = iv[0][2]
for s, e, n, q in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
print("window %.2f ms, %d dispatches, GPU busy (union) %.2f ms = %.1f %%, idle %.2f ms" % (span / 1e6, len(iv), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
per_q = {}
for s, e, n, q in iv:
    per_q[q] = per_q.get(q, 0) + (e - s)
print("per-queue kernel time (ms):", {k: round(v / 1e6, 2) for k, v in per_q.items()})
import collections
by = collections.Counter()
for g, a, b in gaps:
    by[(a, b)] += g
print("idle time by (previous kernel -> next kernel), top 25:")
for (a, b), g in by.most_common(25):
    print("  %8.1f us  %s -> %s" % (g / 1e3, a, b))
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1e3) // 5 * 5, 50)] += g
print("idle by gap size (us bucket -> total us):", {k: round(v / 1e3) for k, v in sorted(hist.items())})
