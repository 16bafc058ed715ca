import sys
sys.path.insert(0, "/root/repo")
from smc_jl_amd import Engine
from tests import models
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
e = Engine(n, 10, seed=1, max_stages=400, store_history=True)
e.set_model(models.gauss_spec(10))
for rep in range(3):
    e.init_from_prior()
    try:
        r = e.run(use_fixed_schedule=False, tempering_target=0.97, use_graph=2)
        print({k: r[k] for k in ("n_stages", "resamples", "logmdd", "seconds", "n_segments", "segment_stages", "kernel_ms_segments", "kernel_ms_mutate", "n_mutate_launches")},
              "us per segment stage %.2f" % (1e3 * r["kernel_ms_segments"] / max(r["segment_stages"], 1)))
    except Exception as ex:
        print("ERR", ex)
