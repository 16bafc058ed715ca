#!/usr/bin/env python
"""examples/regression_model/estimate_regression.jl of the reference, through the Python mirror of `smc(...)` on one MI355X.
Same model (y = α + β x + ε, σ² = 1, priors N(0, 10²)), same committed data, same keyword defaults."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smc_jl_amd as S  # noqa: E402

data = np.load(os.path.join(ROOT, "tests", "golden", "reg_data.npz"))["data"]          # 100 x 2 = [y X]
parameters = [S.parameter("α1", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False),
              S.parameter("β1", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False)]
cloud, w, W = S.smc(S.LinReg(1.0), parameters, data, n_parts=int(sys.argv[1]) if len(sys.argv) > 1 else 1000,
                    use_fixed_schedule=True, seed=1793, verbose="low", savepath=None)
print("posterior mean", S.weighted_mean(cloud), " (exact 1.00018685, 0.99936133)")
print("posterior cov\n", S.weighted_cov(cloud), "\n (exact [[0.03338489, -0.05207003], [-0.05207003, 0.11593692]])")
print("log-MDD %.6f (exact -99.889011)" % cloud.logmdd)
