#!/usr/bin/env python3
"""Where is the GPU idle inside the last full step of a rocprofv3 rocpd database?

    python tools/rocpd_gaps.py x_results.db [min_gap_us=20] [marker]
Lists every interval of the last full step (marker to marker, default pack_conv3x3_many) in which NO dispatch is running and
that is longer than min_gap_us, with the dispatch that ended before it and the one that starts after it, then the sum of all
idle intervals by the kernel that FOLLOWS them, and the idle time between the step's last dispatch and the next step's first.
"""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n)[:70]


def main():
    con = sqlite3.connect(sys.argv[1])
    min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
    marker = sys.argv[3] if len(sys.argv) > 3 else "pack_conv3x3_many"
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3:
        print("fewer than three marker launches")
        return
    a, b = marks[-3], marks[-2]
    step = rows[a:b + 1]  # includes the next step's first dispatch: the step-boundary gap is the last one listed
    t0 = step[0][1]
    busy_end, prev = step[0][2], step[0][0]
    idle_by_next = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    print(f"step of {len(step) - 1} dispatches, {(step[-1][1] - t0) / 1e6:.2f} ms marker to marker")
    for n, s, e in step[1:]:
        if s > busy_end:
            gap = s - busy_end
            total += gap
            k = idle_by_next[short(n)]
            k[0] += 1
            k[1] += gap
            if gap >= min_gap:
                print(f"  t = {(busy_end - t0) / 1e6:8.3f} ms  idle {gap / 1e3:8.1f} us   after {short(prev)[:48]:48s} before {short(n)[:48]}")
        if e > busy_end:
            busy_end, prev = e, n
    print(f"idle in the step (boundary included): {total / 1e6:.3f} ms; by the dispatch that follows the gap:")
    for n, (c, t) in sorted(idle_by_next.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {n:70s} {c:5d} gaps {t / 1e3:9.1f} us")


if __name__ == "__main__":
    main()
