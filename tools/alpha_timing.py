"""Config 2 with a proposal mixture (alpha = 0.9): time per run and per stage (development)."""
import sys
sys.path.insert(0, "/root/repo")
from tests import models
from smc_jl_amd import Engine
for alpha in (1.0, 0.9):
    for n in (100000, 1000000):
        e = Engine(n, 10, seed=1, max_stages=1500, store_history=False)
        e.set_model(models.gauss_spec(10)); e.init_from_prior()
        P0 = e.download_cloud()
        best = 1e9
        for rep in range(3):
            e.upload_cloud(P0)
            r = e.run(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, alpha=alpha)
            best = min(best, r["seconds"])
        print("alpha", alpha, "n", n, "stages", r["n_stages"], "ms %.2f" % (best * 1e3), "us/stage %.1f" % (best * 1e6 / (r["n_stages"] - 1)), "logmdd", r["logmdd"])
        e.close()
