#!/usr/bin/env python3
"""Per-dispatch durations of one kernel over a run (rocprofv3 rocpd database): start offset [s], duration [ms]."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels") or "kernel_dispatch" in t]
rows = None
for t in ("kernels",) + tuple(kd):
    try:
        rows = db.execute("select name, start, end from %s order by start" % t).fetchall()
        break
    except sqlite3.Error:
        continue
if rows is None:
    print("tables:", tabs)
    sys.exit(1)
rows = [r for r in rows if pat in r[0]]
t0 = rows[0][1]
for i, (n, s, e) in enumerate(rows):
    if i % max(1, len(rows) // 25) == 0 or i == len(rows) - 1:
        print("%4d  t=%7.3f s  %.3f ms" % (i, (s - t0) / 1e9, (e - s) / 1e6))
