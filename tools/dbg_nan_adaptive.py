"""development: an adaptive run on a cloud with one NaN log-likelihood (must end with an error, not hang)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smc_jl_amd import Engine
from smc_jl_amd.host._lib import SMCMIError
from tests import models
n, d = 20480, 10
e = Engine(n, d, seed=3, max_stages=400, store_history=True)
e.set_model(models.gauss_spec(d)); e.init_from_prior()
P = e.download_cloud()
P[123, d] = np.nan
e.upload_cloud(P)
t0 = time.time()
try:
    r = e.run(use_fixed_schedule=False, tempering_target=0.95)
    print("finished", r["n_stages"], r["logmdd"], time.time() - t0)
except SMCMIError as ex:
    print("error", ex, time.time() - t0)
