#!/usr/bin/env python3
"""Compact per-kernel summary (qrl:: kernels only) from a rocprofv3 --kernel-trace --stats rocpd
database (*_results.db).  Usage: prof_summary.py <results.db> [label]  -> markdown table on stdout."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = db.execute("select name, total_calls, total_duration, average from top_kernels").fetchall()
    rows = [r for r in rows if "qrl::" in r[0]]
    tot = sum(r[2] for r in rows)
    print("## %s\n" % label)
    print("| kernel | calls | total us | avg us | % of qrl kernels |")
    print("|---|---|---|---|---|")
    for name, calls, total, avg in sorted(rows, key=lambda r: -r[2]):
        short = name.split("(")[0].replace("void ", "")
        print("| %s | %d | %.1f | %.1f | %.1f |" % (short, calls, total, avg, 100.0 * total / tot))
    print()


if __name__ == "__main__":
    main()
