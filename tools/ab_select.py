import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import models
from smc_jl_amd import Engine
which = sys.argv[1]
if which == "capm":
    spec, d, kw, n = models.capm_spec(), 9, dict(use_fixed_schedule=True, n_phi=300, n_mh_steps=int(sys.argv[2])), 20000
else:
    spec, d, kw, n = models.regression_spec(), 2, dict(use_fixed_schedule=True, n_phi=300), 20000
e = Engine(n, d, seed=3, max_stages=300)
e.set_model(spec); e.init_from_prior()
r = e.run(**kw)
rec = e.stage_records(r["n_stages"])
print(which, os.environ.get("SMCMI_NO_SELECT_PREDICT"), repr(r["logmdd"]), r["resamples"], r["select_stalls"], repr(float(rec["ess"][50])), repr(float(rec["ess"][-1])), sorted(rec.keys()))
