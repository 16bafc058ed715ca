import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smc_jl_amd as S
from tests import models
data = models.regression_spec()["lik"][2]
old = np.ascontiguousarray(data[:60])
pars = [S.parameter("a", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False),
        S.parameter("b", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False)]
cloud, _, _ = S.smc(S.LinReg(1.0), pars, old, n_parts=4000, n_phi=60, use_fixed_schedule=True, seed=5, verbose="none")

try:
    c, w, W = S.smc(S.LinReg(1.0), pars, data, old_data=old, old_cloud=cloud, n_parts=4000, n_phi=40, use_fixed_schedule=False,
                    tempering_target=0.9, seed=11, verbose="none", tempered_update_prior_weight=0.5, log_prob_old_data=-3.0)
    print("ok", c.stage_index)
except Exception as e:
    print("ERR", e)
