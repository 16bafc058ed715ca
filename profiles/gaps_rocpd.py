#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd database): for every ordered pair (kernel A ends, kernel B starts
next on the device) the median / mean gap, for the pairs that occur often.  A stage that is a chain of launches pays this per boundary.
usage: python profiles/gaps_rocpd.py <results.db> [min count]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
mincount = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pairs = {}
for (na, sa, ea), (nb, sb, eb) in zip(rows, rows[1:]):
    key = (na.split("(")[0].replace("void ", "").replace("smcmi::", "")[:34], nb.split("(")[0].replace("void ", "").replace("smcmi::", "")[:34])
    pairs.setdefault(key, []).append((sb - ea) / 1e3)
print("%-36s -> %-36s %7s %9s %9s %9s" % ("kernel A", "kernel B", "count", "median_us", "mean_us", "p90_us"))
for key, v in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
    if len(v) < mincount:
        continue
    v.sort()
    print("%-36s -> %-36s %7d %9.2f %9.2f %9.2f" % (key[0], key[1], len(v), v[len(v) // 2], sum(v) / len(v), v[int(0.9 * len(v))]))
