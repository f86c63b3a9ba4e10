#!/usr/bin/env python3
"""Anatomy of the LAST full training step in a rocprofv3 rocpd database: where the wall time of the step goes.

    python tools/rocpd_step.py x_results.db [marker]
A step runs from one launch of `marker` (default pack_conv3x3_many: the step's weight packing) to the next.  Prints, for the
forward part (up to the L1 loss kernel) and the backward part separately: span, union of busy time (dispatches overlap when
the weight gradients run on the side stream), idle time, and per kernel name: launches, mean duration and mean PITCH -- the
time from this dispatch's start to the start of the next dispatch that begins after it ends (what the launch costs the
chain, gaps and tails included).
"""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")][:78] if "(" in n else n[:78]


def part(rows, title):
    if not rows:
        return
    span = rows[-1][2] - rows[0][1]
    ev = sorted([(s, 1) for _, s, e in rows] + [(e, -1) for _, s, e in rows])
    busy, depth, last, two = 0, 0, ev[0][0], 0
    for t, d in ev:
        if depth > 0:
            busy += t - last
        if depth > 1:
            two += t - last
        depth += d
        last = t
    print(f"{title}: {len(rows)} dispatches, span {span / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle "
          f"{(span - busy) / 1e6:.2f} ms, two or more dispatches at once {two / 1e6:.2f} ms")
    acc = collections.defaultdict(lambda: [0, 0, 0, 0])
    starts = [r[1] for r in rows]
    import bisect
    for i, (n, s, e) in enumerate(rows):
        j = bisect.bisect_left(starts, e, i + 1)
        nxt = rows[j][1] if j < len(rows) else e
        a = acc[short(n)]
        a[0] += 1
        a[1] += e - s
        a[2] += nxt - s
        a[3] += max(0, nxt - e)
    print(f"  {'kernel':78s} {'n':>5s} {'dur us':>8s} {'pitch us':>8s} {'gap us':>7s} {'sum ms':>7s}")
    for n, (c, d, p, g) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {n:78s} {c:5d} {d / c / 1e3:8.1f} {p / c / 1e3:8.1f} {g / c / 1e3:7.2f} {d / 1e6:7.2f}")


def main():
    con = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "pack_conv3x3_many"
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        part(rows, "all dispatches")
        return
    step = rows[marks[-2]:marks[-1]]
    print(f"last full step: span {(step[-1][2] - step[0][1]) / 1e6:.2f} ms")
    cut = next((i for i, r in enumerate(step) if "l1_partial" in r[0]), None)
    if cut is None:
        part(step, "step")
    else:
        part(step[:cut], "forward")
        part(step[cut:], "backward + update")


if __name__ == "__main__":
    main()
