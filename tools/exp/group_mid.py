"""development: a cloud of 131 072 < N <= 254 000 particles on ONE GPU as one handle (engine 1) against two in-process handles (sharded
segments on disjoint CUs): config 4 (CAPM, 200 000, fixed schedule, 3 MH steps) and the 10-dim Gaussian at 200 000 / 250 000."""
import sys, time, os
sys.path.insert(0, ".")
from smc_jl_amd import Engine, run_group
from smc_jl_amd.host import workloads as W
cases = [("capm", W.capm_spec(), 9, 200_000, dict(use_fixed_schedule=True, n_phi=300, lam=2.0, n_mh_steps=3)),
         ("gauss10", W.gauss_spec(10), 10, 200_000, dict(use_fixed_schedule=False, tempering_target=0.97)),
         ("gauss10", W.gauss_spec(10), 10, 250_000, dict(use_fixed_schedule=False, tempering_target=0.97))]
for name, spec, d, n, kw in cases:
    for world in (1, 2):
        engs = [Engine(n, d, seed=13, max_stages=1500, store_history=False, n_local=n // world, gid0=r * (n // world)) for r in range(world)]
        for e in engs: e.set_model(spec)
        best = 1e9
        for rep in range(3):
            for e in engs: e.init_from_prior()
            t0 = time.perf_counter()
            res = run_group(engs, **kw) if world > 1 else engs[0].run(**kw)
            best = min(best, time.perf_counter() - t0)
        print("%s n=%d handles %d: %.2f ms, %d stages, %d resamples, %.1f us/stage, segments %s, logmdd %.6f" % (
            name, n, world, 1e3 * best, res["n_stages"], res["resamples"], 1e6 * best / (res["n_stages"] - 1), res.get("n_segments"), res["logmdd"]), flush=True)
        for e in engs: e.close()
