"""development: the reference's own sizes (1 000 - 5 000 particles, fixed schedule of 300 stages) - time per run, launches, stalls"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smc_jl_amd import Engine
from smc_jl_amd.host import workloads as W
for name, spec, d, n, kw in (("regression", W.regression_spec(), 2, 1000, dict(use_fixed_schedule=True, n_phi=300, lam=2.0)),
                             ("gauss10", W.gauss_spec(10), 10, 1000, dict(use_fixed_schedule=True, n_phi=300, lam=2.0)),
                             ("gauss10", W.gauss_spec(10), 10, 5000, dict(use_fixed_schedule=True, n_phi=300, lam=2.0)),
                             ("gauss10", W.gauss_spec(10), 10, 5000, dict(use_fixed_schedule=False, tempering_target=0.97))):
    e = Engine(n, d, seed=3, max_stages=1500, store_history=False)
    e.set_model(spec)
    best = 1e9
    for rep in range(4):
        e.init_from_prior()
        t0 = time.perf_counter()
        r = e.run(**kw)
        best = min(best, time.perf_counter() - t0)
    print("%s n=%d %s: %.3f ms, %d stages, %d resamples, %.1f us/stage, segment launches %d, stalls %s, logmdd %.6f" % (
        name, n, "fixed" if kw.get("use_fixed_schedule") else "adaptive", 1e3 * best, r["n_stages"], r["resamples"], 1e6 * best / (r["n_stages"] - 1), r["n_segments"],
        [r["solver_stalls"], r["select_stalls"], r["spec_stalls"]], r["logmdd"]))
    e.close()
