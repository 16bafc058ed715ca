"""CPU-baseline variants of the oracle on config 2 with their phase times (ORC_PROFILE=1): python tools/cpu_variants.py [threads ...]"""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ["ORC_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from tests import models
spec = models.gauss_spec(10); m = models.oracle_model(spec)
P0 = orc.initial_draw(m, 100000, seed=1)
kw = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, lam=2.1)
cores = os.cpu_count()
r = orc.smc_run(m, P0, seed=1, n_threads=cores, history=False, max_stages=1500, variant=1, **kw)
print("faithful", cores, round(r["seconds"], 3))
for t in [int(a) for a in sys.argv[1:]] or [0]:
    if t:
        os.environ["ORC_OPT_THREADS"] = str(t)
    r = orc.smc_run(m, P0, seed=1, n_threads=cores, history=False, max_stages=1500, variant=2, **kw)
    print("optimised cap", t, round(r["seconds"], 3), r["logmdd"])
