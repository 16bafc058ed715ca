#!/usr/bin/env python
"""Per-kernel HBM traffic and VALU utilisation from separate rocprofv3 PMC passes (each collected with --kernel-trace only, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).

  traffic: FETCH_SIZE (x2: on gfx950 it reports half the bytes of a wide coalesced read) + WRITE_SIZE, KiB -> bytes per launch.
           Calibration on known byte counts of this code base's own 8-byte-per-lane column accesses (N = 100000, round 1):
           a pass reading 3 columns = 2343.75 KiB: counter 1233.25 -> x1.90; writing 2 columns = 1562.5 KiB: counter 1568.8 -> x1.00;
           hipMemcpy D2D of 11718.75 KiB: WRITE 11764.5 (x1.00), FETCH 5886.5 (x1.99).
  valu:    SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD spends issuing VALU instructions, summed over the chip; GRBM_GUI_ACTIVE
           counts the launch's active cycles summed over the 8 XCDs (calibration: 16 852 030 for an 873.28 µs launch = 8 x 2.41 GHz;
           SQ_BUSY_CYCLES likewise sums 32 shader engines).  frac = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
           = share of the launch's SIMD-cycles spent issuing VALU instructions (the bound of the mutation kernel: ~30 FP64 flop per
           byte); for launches of a few tens of µs GRBM_GUI_ACTIVE includes the ramp around the kernel, so the share is a lower
           bound there.  SQ_INSTS_VALU / SQ_WAVES = VALU instructions per wavefront.

usage: python profiles/pmc_extract.py <fetch.db> <write.db> <n_particles> [<sq.db>] > profiles/rNN_pmc_<workload>_n<N>.json"""
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
    (resample stages) do not move a median."""
    pos = sorted(v for v in vals if v > 0)
    if not pos:
        return 0.0, 0, len(vals)
    med = pos[len(pos) // 2]
    act = sorted(v for v in vals if v > 0.5 * med)
    return act[len(act) // 2], len(act), len(vals)


fetch, write, n = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
sq = sys.argv[4] if len(sys.argv) > 4 else None
res = {"n_particles": n, "fetch_correction": 2.0, "write_correction": 1.0, "unit": "bytes per active launch", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, nf, tf = active_mean(fetch.get(k, [0.0]))
    w, nw, tw = active_mean(write.get(k, [0.0]))
    rd, wr = 2.0 * f * 1024.0, w * 1024.0
    # (sums over ALL launches as well: a persistent segment launch - k3_segment - covers a varying number of stages, so its typical
    # launch means little; bench.py divides the sum by the launches for the traffic of an average launch)
    srd, swr = 2.0 * sum(fetch.get(k, [0.0])) * 1024.0, sum(write.get(k, [0.0])) * 1024.0
    res["kernels"][k.split("(")[0]] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr,
                                       "bytes_per_particle": (rd + wr) / n, "active_launches": nf, "launches": tf,
                                       "sum_read_bytes": srd, "sum_write_bytes": swr, "sum_total_bytes": srd + swr}
if sq:
    cnt = {c: per_kernel(sq, c) for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")}
    for k in sorted(cnt["SQ_ACTIVE_INST_VALU"]):
        name = k.split("(")[0]
        m = {c: active_mean(v.get(k, [0.0]))[0] for c, v in cnt.items()}
        frac = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0) if m["GRBM_GUI_ACTIVE"] else None
        res["kernels"].setdefault(name, {})["valu"] = {
            "frac": frac, "unit": "share of SIMD cycles issuing VALU: 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
            "SQ_ACTIVE_INST_VALU": m["SQ_ACTIVE_INST_VALU"], "SQ_INSTS_VALU": m["SQ_INSTS_VALU"], "SQ_WAVES": m["SQ_WAVES"],
            "GRBM_GUI_ACTIVE": m["GRBM_GUI_ACTIVE"], "SQ_BUSY_CYCLES": m["SQ_BUSY_CYCLES"], "SQ_WAVE_CYCLES": m["SQ_WAVE_CYCLES"],
            "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"] if m["SQ_WAVES"] else None}
print(json.dumps(res, indent=1))
