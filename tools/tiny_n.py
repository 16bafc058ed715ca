"""Ragged / tiny clouds: device vs oracle for n not a multiple of anything (development)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import models
from smc_jl_amd import Engine
from oracle import oracle as orc
for n in (33, 64, 65, 127, 257, 1000, 1023, 100003):
    for fixed in (True, False):
        spec = models.regression_spec()
        e = Engine(n, 2, seed=9, max_stages=600)
        e.set_model(spec); e.init_from_prior()
        P0 = e.download_cloud()
        kw = dict(use_fixed_schedule=fixed, n_phi=40, tempering_target=0.9, n_blocks=2)
        try:
            r = e.run(**kw); err = None
        except Exception as ex:
            r, err = None, str(ex)[:50]
        rec = e.stage_records(r["n_stages"]) if r else None
        e.close()
        try:
            ro = orc.smc_run(models.oracle_model(spec), P0, seed=9, n_threads=4, max_stages=600, **kw); oerr = None
        except Exception as ex:
            ro, oerr = None, str(ex)[:50]
        if r and ro:
            print(n, fixed, "stages", r["n_stages"], ro["n_stages"], "rs", r["resamples"], ro["resamples"], "dlogmdd %.2e" % abs(r["logmdd"] - ro["logmdd"]),
                  "ess err %.1e" % np.max(np.abs(rec["ess"] - ro["ess"]) / ro["ess"]) if r["n_stages"] == ro["n_stages"] else "STAGES DIFFER")
        else:
            print(n, fixed, "device:", err, "| oracle:", oerr)
