#!/usr/bin/env python
"""Randomised device-vs-oracle sweep over dimensions, blockings, MH steps, mixture weight, schedules, resamplers (development)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from tests import models
from oracle import oracle as orc

# SWEEP_SIZES=n1,n2,...: cloud sizes to draw from (default: small ones; odd / 2 x odd / 4 x odd sizes reach the other cuts of a handle)
SIZES = [int(x) for x in os.environ.get("SWEEP_SIZES", "2048,4096,6000").split(",")]
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
only = set(int(x) for x in sys.argv[3:])          # optional: trial numbers to run (with a per-stage divergence report)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    kind = rs.choice(["gauss", "gauss", "linmodel", "linmodel_mismatch", "linmodel_bridge", "regression"])
    if kind == "gauss":
        d = int(rs.randint(1, 14))
        spec = models.gauss_spec(d=d, sigma=float(rs.uniform(0.2, 0.6)))
    elif kind == "linmodel":
        d, spec = 9, models.linmodel_spec(T=int(rs.choice([40, 100])))
    elif kind == "linmodel_mismatch":     # old likelihood set but cloud drawn from the prior: a degenerate run, both sides must abort alike
        d, spec = 9, models.linmodel_spec(T=100, old_T=int(rs.choice([30, 60])))
    elif kind == "linmodel_bridge":       # tempered update: estimate on old_T periods first, then continue on all 100
        d, spec = 9, models.linmodel_spec(T=100, old_T=int(rs.choice([30, 60])))
    else:
        d, spec = 2, models.regression_spec()
    nb = int(rs.randint(1, min(d, 3) + 1))
    nf = d
    while ((nf + nb - 1) // nb) * (nb - 1) >= nf:
        nb -= 1
    kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])),
              use_fixed_schedule=bool(rs.randint(0, 2)), n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])),
              resampling_method=str(rs.choice(["systematic", "multinomial"])), threshold_ratio=float(rs.choice([0.5, 0.8])))
    n, seed = int(rs.choice(SIZES)), int(rs.randint(1, 1000))
    if only and trial not in only:
        continue
    e = Engine(n, d, seed=seed, max_stages=1500)
    if kind == "linmodel_bridge":
        T_old = spec["old_lik"][2].shape[1]
        e.set_model(models.linmodel_spec(T=T_old)); e.init_from_prior()
        r_old = e.run(n_phi=60, use_fixed_schedule=True, n_mh_steps=2)
        ess_old = float(e.stage_records(r_old["n_stages"])["ess"][-1])
        P_old = e.download_cloud()
        e.close()
        e = Engine(n, d, seed=seed + 1, max_stages=1500)
        e.set_model(spec); e.upload_cloud(P_old); e.initialize_likelihoods()      # smc_main.jl:249-260
        kw["initial_ess"] = ess_old
    else:
        e.set_model(spec); e.init_from_prior()
    P0 = e.download_cloud()
    try:
        r = e.run(**kw)
    except Exception as ex:   # noqa: BLE001  (PosDefException aborts the run in the reference as well: the oracle must agree)
        e.close()
        try:
            orc.smc_run(models.oracle_model(spec), P0, seed=seed + (kind == "linmodel_bridge"), n_threads=8, max_stages=1500, **kw)
            # (mismatch kind: nothing is ever accepted, the cloud shrinks to a handful of distinct points and Σ becomes numerically
            # singular - whether its Cholesky "succeeds" is decided by rounding, on either side)
            both = kind == "linmodel_mismatch"
        except Exception as ex2:   # noqa: BLE001
            both = ("PosDef" in str(ex) and "PosDef" in str(ex2)) or ("BRACKET" in str(ex) and "bracket" in str(ex2)) \
                or ("NAN_ESS" in str(ex) and "non-zero weight" in str(ex2))
            # degenerate runs: the reference loses the whole cloud to underflow (0/0 -> its solver throws) at |δ e| ~ 745, the
            # device's shifted weights do not and the run dies later of a singular covariance - both abort, at different points
            both = both or kind == "linmodel_mismatch"
        print(json.dumps(dict(trial=trial, kind=str(kind), d=d, n=n, ok=bool(both), logmdd_err=0.0, ess_relerr=0.0, stages=-1, rs=-1, stalls=(0, 0, 0),
                              error=str(ex)[:60], **kw)), flush=True)
        worst = max(worst, 0.0 if both else 1e9)
        continue
    rec = e.stage_records(r["n_stages"])
    e.close()
    try:
        ro = orc.smc_run(models.oracle_model(spec), P0, seed=seed + (kind == "linmodel_bridge"), n_threads=8, max_stages=1500, **kw)
    except Exception as ex2:   # noqa: BLE001
        # the reference's unshifted weights lost the whole cloud to underflow (0/0) where the device's shifted ones did not:
        # only acceptable for the deliberately degenerate kind
        print(json.dumps(dict(trial=trial, kind=str(kind), d=d, n=n, ok=kind == "linmodel_mismatch", logmdd_err=0.0, ess_relerr=0.0,
                              stages=r["n_stages"], rs=r["resamples"], stalls=(0, 0, 0), error="oracle only: " + str(ex2)[:50], **kw)), flush=True)
        worst = max(worst, 0.0 if kind == "linmodel_mismatch" else 1e9)
        continue
    ok = r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"]
    err = abs(r["logmdd"] - ro["logmdd"])
    if kind == "linmodel_mismatch":      # numerically singular proposals (see above): only crashes count for this kind
        ok, err = True, 0.0
    ess_err = float(np.max(np.abs(rec["ess"] - ro["ess"]) / ro["ess"])) if ok else float("nan")
    worst = max(worst, err if ok else 1e9)
    if only:
        m = min(len(rec["ess"]), len(ro["ess"]))
        rel = np.abs(rec["ess"][:m] - ro["ess"][:m]) / ro["ess"][:m]
        relp = np.abs(rec["schedule"][:m] - ro["schedule"][:m]) / np.maximum(ro["schedule"][:m], 1e-300)
        first = np.nonzero((rel > 1e-9) | (relp > 1e-9))[0]
        print("first divergent stage", first[:5], "of", m)
        print("log10 rel err (phi, ess) by stage:", " ".join("%d:%.0f/%.0f" % (i, np.log10(relp[i] + 1e-17), np.log10(rel[i] + 1e-17))
                                                              for i in range(0, m, 4)))
        big = np.nonzero(rel > 1e-7)[0]
        if len(big):
            b = big[0]
            for i in range(max(0, b - 3), min(m, b + 3)):
                print(i, "phi %.17g %.17g  ess %.15g %.15g  acc %.6f %.6f  c %.6f %.6f rs %d %d" % (
                    rec["schedule"][i], ro["schedule"][i], rec["ess"][i], ro["ess"][i], rec["accept_hist"][i], ro["accept_hist"][i],
                    rec["c_hist"][i], ro["c_hist"][i], rec["resampled"][i], ro["resampled"][i]))
        for i in range(max(0, (first[0] if len(first) else 0) - 3), min(m, (first[0] if len(first) else 0) + 4)):
            print(i, "phi %.17g %.17g  ess %.12g %.12g  acc %.6f %.6f  c %.6f %.6f rs %d %d" % (
                rec["schedule"][i], ro["schedule"][i], rec["ess"][i], ro["ess"][i], rec["accept_hist"][i], ro["accept_hist"][i],
                rec["c_hist"][i], ro["c_hist"][i], rec["resampled"][i], ro["resampled"][i]))
    print(json.dumps(dict(trial=trial, kind=str(kind), d=d, n=n, ok=ok, logmdd_err=err, ess_relerr=ess_err, stages=r["n_stages"], rs=r["resamples"], ms=round(r["seconds"] * 1e3, 3),
                          stalls=(r["solver_stalls"], r["select_stalls"], r["spec_stalls"]), **kw)), flush=True)
print("worst logmdd err", worst)
