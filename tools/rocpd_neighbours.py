#!/usr/bin/env python3
"""Which kernels run right before / after the dispatches whose name contains PATTERN (rocpd SQLite database)?"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = con.execute("select name, start, end from kernels order by start").fetchall()
prev, nxt, dur = collections.Counter(), collections.Counter(), collections.Counter()
for i, (name, s, e) in enumerate(rows):
    if pat in name:
        prev[rows[i - 1][0][:90] if i else "-"] += 1
        nxt[rows[i + 1][0][:90] if i + 1 < len(rows) else "-"] += 1
        dur[(e - s) // 500 * 500] += 1
print("before:", prev.most_common(8))
print("after:", nxt.most_common(8))
print("durations (ns, 500-ns bins):", sorted(dur.items())[:20])
# idle time between consecutive dispatches (gaps below 100 us = inside a step)
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 <= g < 100000]
neg = [g for g in gaps if g < 0]
busy = sum(e - s for _, s, e in rows)
print(f"dispatches {len(rows)}, busy {busy / 1e6:.2f} ms, in-step gaps {sum(small) / 1e6:.2f} ms over {len(small)} "
      f"(mean {sum(small) / max(len(small), 1):.0f} ns), overlapping pairs {len(neg)}")
hist = collections.Counter(min(g // 500 * 500, 10000) for g in small)
print("gap histogram (ns):", sorted(hist.items()))
# the last full step only: from the last pack_conv3x3_many launch to the end
starts = [i for i, r in enumerate(rows) if "pack_conv3x3_many" in r[0]]
if len(starts) >= 2:
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    g = [step[i + 1][1] - step[i][2] for i in range(len(step) - 1)]
    print(f"last full step: {len(step)} dispatches, span {(step[-1][2] - step[0][1]) / 1e6:.2f} ms, busy "
          f"{sum(e - s for _, s, e in step) / 1e6:.2f} ms, gaps {sum(x for x in g if x > 0) / 1e6:.2f} ms")
    big = collections.Counter()
    for i, x in enumerate(g):
        if x > 3000:
            big[(step[i][0][:60], step[i + 1][0][:60])] += 1
    print("gaps > 3 us between:", big.most_common(12))
    hist = collections.Counter(min(max(x, 0) // 500 * 500, 10000) for x in g)
    print("gap histogram (ns):", sorted(hist.items()))
