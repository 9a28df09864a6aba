#!/usr/bin/env python3
"""Print a per-kernel summary (calls, total/avg/min/max ns, %) from a rocprofv3 rocpd sqlite database.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/name.txt]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cur = c.execute("select * from top_kernels")
cols = [d[0] for d in cur.description]
rows = cur.fetchall()
print(" | ".join(cols))
for r in rows:
    print(" | ".join(str(x) for x in r))
