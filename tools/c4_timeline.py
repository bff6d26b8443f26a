#!/usr/bin/env python3
"""Steady-state kernel timeline of the C4 receiver from a rocprofv3 --kernel-trace rocpd database: start offset, duration and queue of the
channelizer / per-channel / symbol-sync kernels of the last PFB-form steps (the form-2 run that follows in bench.py is cut off).
Usage: c4_timeline.py <results.db> [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
import os
every = bool(os.environ.get("QRL_TL_ALL"))   # QRL_TL_ALL=1: the runtime's own kernels too (fills, device-to-device copies)
rows = [(n.split("(")[0].replace("void ", "").replace("qrl::", ""), s, e, qq) for n, s, e, qq in rows if "qrl::" in n or (every and "at::" not in n and "elementwise" not in n)]
cut = next((i for i, r in enumerate(rows) if r[0].startswith("k_decim_mfma")), len(rows))
rows = rows[:cut]
last = [i for i, r in enumerate(rows) if r[0].startswith("k_pfb")][-steps:]
rows = rows[last[0]:]
t0 = rows[0][1]
for n, s, e, qq in rows:
    print("%9.1f us  +%8.1f us  ends %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, qq, n[:50]))
