#!/usr/bin/env python3
"""per-step kernel table from a rocprofv3 results .db of `bench.py`: python tools/prof_steps.py DB [first_step last_step]
Steps are delimited by the k_adamw_ema launches; prints busy time, span and launches per step and the per-kernel split."""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    idx = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
    a = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    b = int(sys.argv[3]) if len(sys.argv) > 3 else min(len(idx) - 1, a + 10)
    n = b - a
    sel = rows[idx[a] + 1: idx[b] + 1]
    agg = collections.defaultdict(lambda: [0, 0])
    gaps = 0
    for i, (name, s, e) in enumerate(sel):
        name = name.split('(')[0][:78]
        agg[name][0] += 1
        agg[name][1] += e - s
        if i:
            gaps += max(0, s - max(r[2] for r in sel[max(0, i - 4):i]))
    tot = sum(v[1] for v in agg.values())
    span = sel[-1][2] - sel[0][1]
    print("steps %d..%d: busy %.3f ms/step, span %.3f ms/step, idle gaps %.3f ms/step, %.1f launches/step"
          % (a, b, tot / n / 1e6, span / n / 1e6, gaps / n / 1e6, len(sel) / n))
    for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 50]:
        print("%-78s %6.1f/step %8.1f us/step  avg %7.1f us" % (name, v[0] / n, v[1] / n / 1e3, v[1] / v[0] / 1e3))


if __name__ == "__main__":
    main()
