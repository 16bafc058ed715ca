"""Worker of tests/test_gpu_sweep.py: the randomised device-vs-oracle sweep in a process of its own, so that the engine-selection
switches (read once per process) can be set per run.  usage: python tests/sweep_worker.py <rng seed> <trials> <d_max>
Prints one JSON line per trial and a final "DONE <n>" line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc          # noqa: E402
from smc_jl_amd import Engine             # noqa: E402
from tests import models                  # noqa: E402

rs = np.random.RandomState(int(sys.argv[1]))
n_trials, d_max = int(sys.argv[2]), int(sys.argv[3])
for trial in range(n_trials):
    d = int(rs.randint(1, d_max + 1))
    spec = models.gauss_spec(d=d, sigma=float(rs.uniform(0.2, 0.6)))
    nb = int(rs.randint(1, min(d, 3) + 1))
    while ((d + nb - 1) // nb) * (nb - 1) >= d:
        nb -= 1
    kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])),
              use_fixed_schedule=bool(rs.randint(0, 2)), n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])),
              resampling_method=str(rs.choice(["systematic", "multinomial"])), threshold_ratio=float(rs.choice([0.5, 0.8])))
    n, seed = int(rs.choice([2048, 4096, 6000])), int(rs.randint(1, 1000))
    e = Engine(n, d, seed=seed, max_stages=1500)
    e.set_model(spec)
    e.init_from_prior()
    P0 = e.download_cloud()
    r = e.run(**kw)
    rec = e.stage_records(r["n_stages"])
    e.close()
    ro = orc.smc_run(models.oracle_model(spec), P0, seed=seed, n_threads=8, max_stages=1500, **kw)
    m = min(len(rec["ess"]), len(ro["ess"]))
    print(json.dumps(dict(trial=trial, d=d, n=n, kw=kw, stages=[r["n_stages"], ro["n_stages"]], resamples=[r["resamples"], ro["resamples"]],
                          logmdd_err=abs(r["logmdd"] - ro["logmdd"]),
                          ess_relerr=float(np.max(np.abs(rec["ess"][:m] - ro["ess"][:m]) / ro["ess"][:m])),
                          phi_relerr=float(np.max(np.abs(rec["schedule"][:m] - ro["schedule"][:m]) / np.maximum(ro["schedule"][:m], 1e-300))))), flush=True)
print("DONE %d" % n_trials)
