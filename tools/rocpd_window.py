#!/usr/bin/env python3
"""A window of consecutive dispatches of the last full step of a rocpd database: start offset, duration, queue, name.

    python tools/rocpd_window.py x_results.db [first=300] [count=60] [marker]
Shows how parallel branches of a replayed hipGraph (the sample lanes) are interleaved on the hardware queues.
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 300
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
marker = sys.argv[4] if len(sys.argv) > 4 else "pack_conv3x3_many"
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
rows = con.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
step = rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows
t0 = step[0][1]
print("columns of the kernels view:", cols)
for n, s, e, q in step[first:first + count]:
    n = n.replace("void ", "")
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:6.1f}  q={q}  {n[:n.index('(')][:70] if '(' in n else n[:70]}")
