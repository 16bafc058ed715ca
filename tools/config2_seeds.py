#!/usr/bin/env python
"""Config 2 of SURVEY §8(d) for seeds {1,2,3} x alpha {1.0, 0.9}: GPU vs CPU-oracle log-MDD at N = 100k (parity at full size)."""
import json
import os
import sys

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # the oracle's idle OpenMP threads must not spin into the next GPU run's launch path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine  # noqa: E402
from tests import models  # noqa: E402
from oracle import oracle as orc  # noqa: E402

spec = models.gauss_spec(10)
m = models.oracle_model(spec)
out = []
for seed in (1, 2, 3):
    for alpha in (1.0, 0.9):
        eng = Engine(100000, 10, seed=seed, max_stages=1500)
        eng.set_model(spec)
        eng.init_from_prior()
        P0 = eng.download_cloud()
        kw = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, lam=2.1, alpha=alpha)
        r = eng.run(**kw)
        mu = eng.moments()[0]
        eng.close()
        ro = orc.smc_run(m, P0, seed=seed, n_threads=os.cpu_count(), history=False, max_stages=1500, **kw)
        out.append(dict(seed=seed, alpha=alpha, n_stages=r["n_stages"], n_stages_cpu=ro["n_stages"], resamples=r["resamples"],
                        logmdd_gpu=r["logmdd"], logmdd_cpu=ro["logmdd"], abs_err=abs(r["logmdd"] - ro["logmdd"]),
                        gpu_seconds=r["seconds"], cpu_seconds=ro["seconds"], mean_err=float(np.max(np.abs(mu - spec["lik"][2].ravel() * 25 / 25.0625)))))
        print(json.dumps(out[-1]), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "r02_config2_seeds.json"), "w"), indent=1)
