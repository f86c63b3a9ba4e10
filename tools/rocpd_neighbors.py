#!/usr/bin/env python3
"""What runs before and after every dispatch of one kernel in a rocprofv3 kernel trace (rocpd SQLite database).

    python tools/rocpd_neighbors.py x_results.db __amd_rocclr_copyBuffer
"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
extra = [c for c in ("grid_size", "workgroup_size", "grid_x", "grid_size_x", "workgroup_size_x", "queue_id", "stream_id") if c in cols]
rows = con.execute(f"select name, start, end{''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
acc, sizes = collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        prev = rows[i - 1][0][:60] if i else "-"
        nxt = rows[i + 1][0][:60] if i + 1 < len(rows) else "-"
        acc[(prev, nxt)] += 1
        sizes[tuple(r[3:])] += 1
print("columns:", extra)
for k, n in sizes.most_common(12):
    print(n, k)
for (a, b), n in acc.most_common(25):
    print(n, "after", a, "| before", b)
if len(sys.argv) > 3:  # dump the dispatches around the N-th occurrence of a second pattern
    pat2, nth = sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 3
    hits = [i for i, r in enumerate(rows) if pat2 in r[0]]
    if len(hits) > nth:
        i0 = hits[nth]
        t0 = rows[i0][1]
        for r in rows[max(0, i0 - 12):i0 + 30]:
            print(f"{(r[1] - t0) / 1e3:10.1f} us  {(r[2] - r[1]) / 1e3:7.1f} us  {r[0][:70]}  {r[3:]}")
