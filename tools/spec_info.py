"""Stall / pass counters of the adaptive config-2 run for a list of phi_rtol values (development)."""
import sys
sys.path.insert(0, "/root/repo")
from tests import models
from smc_jl_amd import Engine
rtols = [float(x) for x in sys.argv[1:]] or [0.0]
for n in (100000,):
    for rt in rtols:
        e = Engine(n, 10, seed=1, max_stages=1500, store_history=False)
        e.set_model(models.gauss_spec(10)); e.init_from_prior()
        r = e.run(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, phi_rtol=rt)
        print(n, rt, {k: r[k] for k in ("n_stages", "resamples", "solver_passes", "solver_stalls", "select_stalls", "spec_stalls", "seconds", "logmdd")})
        e.close()
