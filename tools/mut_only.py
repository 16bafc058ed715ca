import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from tests import models
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
which = int(sys.argv[2]) if len(sys.argv) > 2 else 9
e = Engine(n, 10, seed=1, max_stages=8, store_history=True)
e.set_model(models.gauss_spec()); e.init_from_prior()
print(e.time_kernel(which, 50))
