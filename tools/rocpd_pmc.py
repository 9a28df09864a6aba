#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from a rocprofv3 rocpd sqlite database (--pmc run).
Usage: python tools/rocpd_pmc.py gpurun_out/pmc/x_results.db"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cur = c.execute("select * from counters_collection limit 1")
cols = [d[0] for d in cur.description]
ki = cols.index("kernel_name") if "kernel_name" in cols else None
rows = c.execute("select * from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    d = dict(zip(cols, r))
    agg[d.get("kernel_name", "?")[:60]][d["counter_name"]].append(d["value"])
for k, cs in agg.items():
    print(k)
    for name, vals in sorted(cs.items()):
        print(f"    {name:28s} avg {sum(vals)/len(vals):18.1f}  n={len(vals)}")
