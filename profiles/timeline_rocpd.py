#!/usr/bin/env python
"""The device timeline of the LAST run of a rocprofv3 kernel trace (rocpd database): every kernel and memory operation from the last
`k_init_prior` on, with its start (µs from the run's first kernel), duration and the idle gap in front of it; k3_segment launches carry
what share of the run they are.  Development aid: where a run's time goes OUTSIDE its stage kernels.
usage: python profiles/timeline_rocpd.py <results.db> [first_kernel_substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_init_prior"
first = max(i for i, r in enumerate(rows) if mark in r[0])
rows = rows[first:]
t0 = rows[0][1]
prev_end = t0
tot = {}
for name, s, e in rows:
    short = name.split("(")[0].replace("void ", "").replace("smcmi::", "")[:40]
    print("%10.2f  gap %8.2f  dur %9.2f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short))
    a = tot.setdefault(short, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(0.0, (s - prev_end) / 1e3)
    prev_end = max(prev_end, e)
print("run span %.2f us" % ((prev_end - t0) / 1e3))
print("%-42s %6s %10s %14s" % ("kernel", "count", "busy_us", "gap_before_us"))
for k, a in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-42s %6d %10.2f %14.2f" % (k, a[0], a[1], a[2]))
