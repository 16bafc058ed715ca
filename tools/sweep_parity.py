#!/usr/bin/env python
"""Randomised device-vs-oracle sweep over dimensions, blockings, MH steps, mixture weight, schedules, resamplers (development)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from tests import models
from oracle import oracle as orc

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    d = int(rs.randint(1, 14))
    spec = models.gauss_spec(d=d, sigma=float(rs.uniform(0.2, 0.6)))
    nb = int(rs.randint(1, min(d, 3) + 1))
    nf = d
    while ((nf + nb - 1) // nb) * (nb - 1) >= nf:
        nb -= 1
    kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])),
              use_fixed_schedule=bool(rs.randint(0, 2)), n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])),
              resampling_method=str(rs.choice(["systematic", "multinomial"])), threshold_ratio=float(rs.choice([0.5, 0.8])))
    n, seed = int(rs.choice([2048, 4096, 6000])), int(rs.randint(1, 1000))
    e = Engine(n, d, seed=seed, max_stages=1500)
    e.set_model(spec); e.init_from_prior()
    P0 = e.download_cloud()
    r = e.run(**kw)
    rec = e.stage_records(r["n_stages"])
    e.close()
    ro = orc.smc_run(models.oracle_model(spec), P0, seed=seed, n_threads=8, max_stages=1500, **kw)
    ok = r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"]
    err = abs(r["logmdd"] - ro["logmdd"])
    ess_err = float(np.max(np.abs(rec["ess"] - ro["ess"]) / ro["ess"])) if ok else float("nan")
    worst = max(worst, err if ok else 1e9)
    print(json.dumps(dict(trial=trial, d=d, n=n, ok=ok, logmdd_err=err, ess_relerr=ess_err, stages=r["n_stages"], rs=r["resamples"],
                          stalls=(r["solver_stalls"], r["select_stalls"]), **kw)), flush=True)
print("worst logmdd err", worst)
