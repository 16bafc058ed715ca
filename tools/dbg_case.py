"""Replay one sweep trial (seed, trial) and print the device and oracle ESS paths side by side (development)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from tests import models
from oracle import oracle as orc
sd, want = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(sd)
for trial in range(want + 1):
    kind = rs.choice(["gauss", "gauss", "linmodel", "linmodel_mismatch", "linmodel_bridge", "regression"])
    if kind == "gauss":
        d = int(rs.randint(1, 14)); spec = models.gauss_spec(d=d, sigma=float(rs.uniform(0.2, 0.6)))
    elif kind == "linmodel":
        d, spec = 9, models.linmodel_spec(T=int(rs.choice([40, 100])))
    elif kind in ("linmodel_mismatch", "linmodel_bridge"):
        d, spec = 9, models.linmodel_spec(T=100, old_T=int(rs.choice([30, 60])))
    else:
        d, spec = 2, models.regression_spec()
    nb = int(rs.randint(1, min(d, 3) + 1))
    while ((d + nb - 1) // nb) * (nb - 1) >= d: nb -= 1
    kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])),
              use_fixed_schedule=bool(rs.randint(0, 2)), n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])),
              resampling_method=str(rs.choice(["systematic", "multinomial"])), threshold_ratio=float(rs.choice([0.5, 0.8])))
    n, seed = int(rs.choice([2048, 4096, 6000])), int(rs.randint(1, 1000))
print(kind, n, seed, kw)
e = Engine(n, d, seed=seed, max_stages=1500)
e.set_model(spec); e.init_from_prior()
P0 = e.download_cloud()
try:
    r = e.run(**kw); ns = r["n_stages"]
except Exception as ex:
    print("device:", ex); ns = int(os.environ.get("NS", "20"))
rec = e.stage_records(ns)
ro = orc.smc_run(models.oracle_model(spec), P0, seed=seed, n_threads=8, max_stages=1500, **kw)
m = min(ns, ro["n_stages"])
for i in range(m):
    print(i, "phi %.6e %.6e ess %.3f %.3f acc %.4f %.4f c %.4f %.4f rs %d %d" % (rec["schedule"][i], ro["schedule"][i], rec["ess"][i], ro["ess"][i],
          rec["accept_hist"][i], ro["accept_hist"][i], rec["c_hist"][i], ro["c_hist"][i], rec["resampled"][i], ro["resampled"][i]))
