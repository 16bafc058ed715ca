"""development: the ESS trajectory of config 4 around its resample stages (stage, ESS, resampled, ratio to the stage before) - the decay is
smooth (ratios 0.74 - 0.99), what a host forecast of resample stages could use.  usage (GPU box): python tools/exp/ess_dump.py"""
import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
from tests import models
from tests.test_gpu_parity import make_engine
spec = models.capm_spec()
eng = make_engine(spec, 200000, seed=1, max_stages=400)
eng.init_from_prior()
r = eng.run(use_fixed_schedule=True, n_phi=300, n_mh_steps=3)
rec = eng.stage_records(r["n_stages"])
e = rec["ess"]; rs = rec["resampled"]
for i in range(len(e)):
    if rs[i] or (i+1 < len(e) and rs[i+1]) or (i+2 < len(e) and rs[i+2]) or i < 12:
        print(i+1, round(e[i]), int(rs[i]), round(e[i]/e[i-1],3) if i else 0)
