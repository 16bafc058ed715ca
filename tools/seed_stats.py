"""Config 2 over many seeds: spread of the log-MDD estimate around the exact value, stall counters (development)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import models
from smc_jl_amd import Engine
spec = models.gauss_spec(10)
exact = models.gauss_logmdd(10)
vals, stalls, stages = [], [], []
for seed in range(1, int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 41):
    e = Engine(100000, 10, seed=seed, max_stages=1500, store_history=False)
    e.set_model(spec); e.init_from_prior()
    r = e.run(use_fixed_schedule=False, tempering_target=0.97, n_phi=300)
    e.close()
    vals.append(r["logmdd"]); stalls.append((r["solver_stalls"], r["select_stalls"], r["spec_stalls"])); stages.append(r["n_stages"])
v = np.array(vals)
print("seeds", len(v), "exact", exact, "mean", v.mean(), "sd", v.std(ddof=1), "mean - exact", v.mean() - exact, "se", v.std(ddof=1) / np.sqrt(len(v)))
print("stages", min(stages), max(stages), "stalls (solver, select, spec) totals", np.array(stalls).sum(axis=0))
