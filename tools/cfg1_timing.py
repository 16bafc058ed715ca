import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import models
from smc_jl_amd import Engine
spec = models.regression_spec()
for n in (1000, 5000, 20000):
    e = Engine(n, 2, seed=1793, max_stages=300)
    e.set_model(spec); e.init_from_prior()
    P0 = e.download_cloud()
    best = 1e9
    for rep in range(5):
        e.upload_cloud(P0)
        r = e.run(use_fixed_schedule=True, n_phi=300, use_graph=int(os.environ.get("UG", "0")))
        best = min(best, r["seconds"])
    print(n, "stages", r["n_stages"], "resamples", r["resamples"], "ms %.2f" % (1e3 * best), "us/stage %.1f" % (1e6 * best / 299), "logmdd %.4f" % r["logmdd"])
    e.close()
