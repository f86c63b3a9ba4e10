#!/usr/bin/env python3
"""Per-kernel summary (calls, total, mean, share) of a rocprofv3 rocpd SQLite database -> CSV on stdout.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                   "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, n, tot, avg, lo, hi in rows:
    print(f'"{name}",{n},{tot},{avg:.1f},{100.0 * tot / total:.3f},{lo},{hi}')
