import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import models
from smc_jl_amd import Engine, run_group
spec = models.gauss_spec(10)
n, world = 100000, 2
engs = []
for r in range(world):
    s = Engine(n, 10, seed=1, max_stages=1500, store_history=False, n_local=n // world, gid0=r * (n // world))
    s.set_model(spec); s.init_from_prior(); engs.append(s)
r = run_group(engs, use_fixed_schedule=False, tempering_target=0.97, n_phi=300)
print({k: r[k] for k in ("n_stages", "resamples", "solver_passes", "solver_stalls", "select_stalls", "spec_stalls", "logmdd")})
