#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table.
usage: python profiles/summarize_rocpd.py gpurun_out/<dir>/<name>_results.db > profiles/<name>_kernel_stats.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x*grid_y), max(workgroup_x) "
                      "from kernels group by name order by 4 desc"))
tot = sum(r[3] for r in rows)
print("%-72s %7s %10s %10s %6s %9s %9s %5s %5s %7s %9s %4s" % ("kernel", "calls", "avg_us", "total_ms", "%", "min_us", "max_us", "vgpr", "agpr", "lds", "grid", "wg"))
for r in rows:
    print("%-72s %7d %10.2f %10.2f %6.1f %9.2f %9.2f %5d %5d %7d %9d %4d" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e6, 100 * r[3] / tot, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], r[9], r[10]))
print("total kernel time %.2f ms" % (tot / 1e6))
