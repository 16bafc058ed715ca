#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, with --kernel-trace
only), corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE under-reports coalesced streaming reads by 2x.
Calibration on known byte counts of this code base's own 8-byte-per-lane column accesses (N = 100000):
  k_pass<1,true> reads 3 columns = 2343.75 KiB, counter 1233.25 -> x1.90 (we apply the guide's x2);
  k_pass<1,true> writes 2 columns = 1562.5 KiB, counter 1568.8 -> x1.00; hipMemcpy D2D 11718.75 KiB: WRITE 11764.5 (x1.00), FETCH 5886.5 (x1.99).
usage: python profiles/pmc_extract.py <fetch.db> <write.db> <n_particles> > profiles/rNN_pmc_traffic.json"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        out.setdefault(name, []).append(val)
    return out


def active_mean(vals):
    """Typical (median) value over the launches that did work: early-exit no-ops are dropped; the rare heavier launches
    (resample stages copy the cloud back inside k_moments) do not move a median."""
    pos = sorted(v for v in vals if v > 0)
    if not pos:
        return 0.0, 0, len(vals)
    med = pos[len(pos) // 2]
    act = sorted(v for v in vals if v > 0.5 * med)
    return act[len(act) // 2], len(act), len(vals)


fetch, write, n = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
res = {"n_particles": n, "fetch_correction": 2.0, "write_correction": 1.0, "unit": "bytes per active launch", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, nf, tf = active_mean(fetch.get(k, [0.0]))
    w, nw, tw = active_mean(write.get(k, [0.0]))
    rd, wr = 2.0 * f * 1024.0, w * 1024.0
    res["kernels"][k.split("(")[0]] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr,
                                       "bytes_per_particle": (rd + wr) / n, "active_launches": nf, "launches": tf}
print(json.dumps(res, indent=1))
