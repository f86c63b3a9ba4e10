#!/usr/bin/env python3
"""Last N dispatches before the final adam_flat launch (name, start offset us, duration us, stream/queue if present)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = con.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_flat" in r[0]]
b = idx[-1]
t0 = rows[b - n][1]
for name, s, e in rows[b - n:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name[:80]}")
