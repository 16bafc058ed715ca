"""Config 2 at N = 1e6 (10x the headline size): GPU vs CPU-oracle log-MDD, stage / resample counts, ESS path (one-off, ~6 min of CPU)."""
import json, os, sys
import numpy as np
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from tests import models
from oracle import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spec = models.gauss_spec(10)
eng = Engine(n, 10, seed=1, max_stages=1500, store_history=False)
eng.set_model(spec); eng.init_from_prior()
P0 = eng.download_cloud()
kw = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, lam=2.1)
r = eng.run(**kw)
rec = eng.stage_records(r["n_stages"])
eng.close()
ro = orc.smc_run(models.oracle_model(spec), P0, seed=1, n_threads=os.cpu_count(), history=False, max_stages=1500, **kw)
same = r["n_stages"] == ro["n_stages"]
out = dict(n=n, n_stages=r["n_stages"], n_stages_cpu=ro["n_stages"], resamples=r["resamples"], resamples_cpu=ro["resamples"],
           logmdd_gpu=r["logmdd"], logmdd_cpu=ro["logmdd"], abs_err=abs(r["logmdd"] - ro["logmdd"]),
           max_rel_ess_err=float(np.max(np.abs(rec["ess"] - ro["ess"]) / ro["ess"])) if same else None,
           max_rel_phi_err=float(np.max(np.abs(rec["schedule"][1:] - ro["schedule"][1:]) / ro["schedule"][1:])) if same else None,
           gpu_seconds=r["seconds"], cpu_seconds=ro["seconds"], cpu_threads=os.cpu_count(), logmdd_exact=models.gauss_logmdd(10))
print(json.dumps(out))
