"""development: config 3's cloud (10^6 particles) as 1 / 2 / 4 / 8 in-process handles - the shapes bench.py --gpus N gives every rank - must leave
the same bits (engine 2's canonical order); prints stage counts, log-MDD bits and the time per run."""
import sys, time, hashlib
import numpy as np
sys.path.insert(0, ".")
from smc_jl_amd import Engine, run_group
from tests import models
n, d = 1_000_000, 10
ref = None
for world in (int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8"])):
    engs = []
    for r in range(world):
        e = Engine(n, d, seed=13, max_stages=1500, store_history=False, n_local=n // world, gid0=r * (n // world))
        e.set_model(models.gauss_spec(d)); e.init_from_prior(); engs.append(e)
    kw = dict(use_fixed_schedule=False, tempering_target=0.97)
    t0 = time.perf_counter()
    res = run_group(engs, **kw) if world > 1 else engs[0].run(**kw)
    dt = time.perf_counter() - t0
    key = (res["n_stages"], res["resamples"], float(res["logmdd"]).hex())
    print("world", world, key, "segments", res.get("n_segments"), "%.1f ms" % (1e3 * dt), flush=True)
    if ref is None: ref = key
    elif key != ref: print("  MISMATCH against world 1")
    for e in engs: e.close()
