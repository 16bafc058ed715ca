import sys
sys.path.insert(0, "/root/repo")
from tests import models
from smc_jl_amd import Engine
for n in (100000, 1000000):
    e = Engine(n, 10, seed=1, max_stages=1500, store_history=False)
    e.set_model(models.gauss_spec(10)); e.init_from_prior()
    r = e.run(use_fixed_schedule=False, tempering_target=0.97, n_phi=300)
    print(n, {k: r[k] for k in ("n_stages", "resamples", "solver_passes", "solver_stalls", "select_stalls", "spec_stalls", "seconds", "logmdd")})
    e.close()
