"""Idle time around one kernel in a rocprofv3 --kernel-trace CSV: mean gap between the previous dispatch's end and its start, and between
its end and the next dispatch's start (python3 tools/gap_probe.py trace.csv k_mhsa_bwd)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
before, after, dur, prevn, nextn = [], [], [], {}, {}
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"] and 0 < i < len(rows) - 1:
        before.append(int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
        after.append(int(rows[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"]))
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        prevn[rows[i - 1]["Kernel_Name"][:40]] = prevn.get(rows[i - 1]["Kernel_Name"][:40], 0) + 1
        nextn[rows[i + 1]["Kernel_Name"][:40]] = nextn.get(rows[i + 1]["Kernel_Name"][:40], 0) + 1
n = len(dur)
if n:
    h = n // 2        # second half: steady state
    print("%s: %d dispatches, duration %.1f us, gap before %.1f us, gap after %.1f us" % (pat, n, sum(dur[h:]) / (n - h) / 1e3, sum(before[h:]) / (n - h) / 1e3, sum(after[h:]) / (n - h) / 1e3))
    print(" prev:", prevn, " next:", nextn)
ce = [int(r["Start_Timestamp"]) for r in rows if "k_soft_ce" in r["Kernel_Name"]]
if len(ce) > 3:
    d = [(b - a) / 1e6 for a, b in zip(ce[:-1], ce[1:])]
    print("interval between k_soft_ce dispatches (ms):", " ".join("%.3f" % x for x in d))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print("sum of kernel durations %.2f ms over a span of %.2f ms" % (busy, span))
