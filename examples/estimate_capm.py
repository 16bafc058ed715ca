#!/usr/bin/env python
"""examples/capm_model/estimate_capm.jl of the reference (likelihood as literally written there, quirk Q12), 3 MH steps per
mutation, through the Python mirror of `smc(...)` on one MI355X."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smc_jl_amd as S  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "capm_data.npz"))
parameters = []
for i in range(1, 4):
    parameters += [S.parameter("α%d" % i, 0.0, (-1e5, 1e5), prior=S.Normal(0, 1e3)),
                   S.parameter("β%d" % i, 0.0, (-1e5, 1e5), prior=S.Normal(0, 1e3)),
                   S.parameter("σ%d" % i, 1.0, (1e-5, 1e5), prior=S.Uniform(0, 1e3))]
cloud, w, W = S.smc(S.CapmLiteral(z["market_data"]), parameters, z["lik_data"], n_parts=int(sys.argv[1]) if len(sys.argv) > 1 else 10000,
                    n_mh_steps=3, use_fixed_schedule=True, seed=1793, verbose="high")
