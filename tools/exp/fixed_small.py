"""development: fixed schedules on small clouds (the reference's default mode): time per stage, stalls, segments"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smc_jl_amd import Engine
from tests import models
for n, nphi in ((100000, 300), (100000, 100), (20000, 300)):
    e = Engine(n, 10, seed=3, max_stages=1500, store_history=False)
    e.set_model(models.gauss_spec(10))
    for rep in range(3):
        e.init_from_prior()
        t0 = time.perf_counter()
        r = e.run(use_fixed_schedule=True, n_phi=nphi, lam=2.0)
        dt = time.perf_counter() - t0
    print("fixed n=%d n_phi=%d: %.3f ms, %d stages, %d resamples, %.1f us/stage, segments %d (stages in them %d), stalls %s, logmdd %.6f" % (
        n, nphi, 1e3 * dt, r["n_stages"], r["resamples"], 1e6 * dt / (r["n_stages"] - 1), r["n_segments"], r["segment_stages"],
        [r["solver_stalls"], r["select_stalls"], r["spec_stalls"]], r["logmdd"]))
    e.close()
