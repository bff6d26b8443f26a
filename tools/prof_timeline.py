#!/usr/bin/env python3
"""Kernel timeline (qrl:: kernels) from a rocprofv3 --kernel-trace rocpd database: start offset, duration, queue.
Usage: prof_timeline.py <results.db> [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
maxr = int(sys.argv[2]) if len(sys.argv) > 2 else 80
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
rows = None
for cand in ("kernels",):
    if cand in names:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % cand)]
        q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else ("stream_id" if "stream_id" in cols else "0"))
        rows = db.execute("select name, start, end, %s from %s order by start" % (q, cand)).fetchall()
        break
if rows is None:
    print("tables/views:", names)
    sys.exit(1)
rows = [r for r in rows if "qrl::" in r[0]]
if not rows:
    print("no qrl kernels")
    sys.exit(0)
rows = rows[-maxr:]
t0 = rows[0][1]
for name, s, e, qid in rows:
    short = name.split("(")[0].replace("void ", "").replace("qrl::", "")
    print("%10.1f us  +%9.1f us  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qid, short[:60]))
