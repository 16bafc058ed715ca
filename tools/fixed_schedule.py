"""Fixed schedules (the reference's default: use_fixed_schedule = true, n_Φ = 300, λ = 2.1; src/smc_main.jl:139) on one handle: ms per run and µs per
stage, best of 5 runs, one JSON line per cloud size.  usage: python tools/fixed_schedule.py [n ...]   (SMCMI_SHIFT_LAG=0: exact energy shifts,
two hand-overs per stage - the comparison run)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from smc_jl_amd.host import workloads as models

sizes = [int(a) for a in sys.argv[1:]] or [100_000, 5_000, 1_000]
for n in sizes:
    for alpha in (1.0, 0.9):
        e = Engine(n, 10, seed=3, max_stages=400, store_history=False)
        e.set_model(models.gauss_spec(10))
        best, r = None, None
        for rep in range(6):
            e.init_from_prior()
            e.sync()
            t0 = time.perf_counter()
            r = e.run(use_fixed_schedule=True, n_phi=300, lam=2.1, alpha=alpha)
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best):
                best = dt
        print(json.dumps(dict(workload="gauss10 fixed schedule n_phi=300", n=n, alpha=alpha, shift_lag=os.environ.get("SMCMI_SHIFT_LAG", "1"), ms_per_run=1e3 * best,
                              us_per_stage=1e6 * best / (r["n_stages"] - 1), n_stages=r["n_stages"], resamples=r["resamples"], segments=r["n_segments"],
                              segment_stages=r["segment_stages"], segment_blocks=r["segment_blocks"], select_stalls=r["select_stalls"],
                              shift_fallback_stage=r["shift_fallback_stage"], logmdd=r["logmdd"])))
        e.close()
