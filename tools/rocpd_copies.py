#!/usr/bin/env python3
"""Device copies of a rocprofv3 --memory-copy-trace run, grouped by (direction, size): where a step's copyBuffer launches come from.

    python tools/rocpd_copies.py x_results.db
"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table', 'view')")]
cand = [n for n in names if "memory_cop" in n.lower() or n.lower() == "memory_copies"]
print("tables:", cand)
for t in cand[:3]:
    cols = [r[1] for r in con.execute(f"pragma table_info('{t}')")]
    print(t, cols)
    size_col = next((c for c in cols if c.lower() in ("size", "bytes")), None)
    name_col = next((c for c in cols if c.lower() in ("name", "kind", "direction")), None)
    if not size_col:
        continue
    acc = collections.Counter()
    for row in con.execute(f"select {name_col or 'NULL'}, {size_col} from '{t}'"):
        acc[(row[0], row[1])] += 1
    for (k, sz), n in acc.most_common(40):
        print(n, k, sz)
