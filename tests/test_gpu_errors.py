"""The loop's failure paths on every stage engine.

check_nan_ess (src/helpers.jl:270-305, called at src/smc_main.jl:427-432): a correction that leaves no usable weight ends the run with
the reference's assertion text.  With the adaptive schedule the reference never reaches it - `fzero` throws on the NaN objective first
(src/helpers.jl:49) - so the guard is exercised on fixed schedules; the adaptive counterpart is the solver's bracket error.
PosDefException (src/mutation.jl:81: `MvNormal(θ̄_b, c²Σ_b)` outside the try block) aborts the run: a cloud whose particles coincide has
a zero covariance.  Inside an engine-3 segment the error must come back at once, not after the hand-over time-out.

Engines: 3 = persistent segments (small cloud, default), 2 = two launches per stage (SMCMI_ENGINE3=0), 1 = round-1 pipeline
(SMCMI_ENGINE=1); a 2-shard in-process group runs engine 2's sharded driver."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_W = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from smc_jl_amd import Engine
from smc_jl_amd.host import engine as eng
from smc_jl_amd.host._lib import SMCMIError
from tests import models
cfg = json.loads(%(cfg)r)
n, d = cfg["n"], cfg["d"]
spec = models.gauss_spec(d)
shards = cfg.get("shards", 1)
es = []
for r in range(shards):
    e = Engine(n, d, seed=3, max_stages=400, store_history=True, n_local=n // shards, gid0=r * (n // shards))
    e.set_model(spec); e.init_from_prior()
    es.append(e)
P = [e.download_cloud() for e in es]
out = []
if cfg.get("warm"):                                                   # first-use costs (allocations, residency self-test) out of the timings
    kw = dict(cfg["kw"])
    eng.run_group(es, **kw) if shards > 1 else es[0].run(**kw)
for case in cfg["cases"]:
    for e, P0 in zip(es, P):
        Q = P0.copy()
        if case == "nan_loglh":
            if e.gid0 == 0: Q[123, d] = np.nan                         # one particle's log-likelihood is NaN
        elif case == "all_minus_inf":
            Q[:, d] = -np.inf                                          # no particle has a usable likelihood
        elif case == "degenerate":
            Q[:, :] = P[0][7, :][None, :]; Q[:, d + 4] = 1.0            # every particle the same point: Σ = 0
        e.upload_cloud(Q)
    t0 = time.time()
    try:
        kw = dict(cfg["kw"])
        if shards > 1:
            r = eng.run_group(es, **kw)
        else:
            r = es[0].run(**kw)
        out.append(dict(case=case, code=0, msg="", seconds=time.time() - t0, n_stages=r["n_stages"], segments=r["n_segments"]))
    except SMCMIError as ex:
        out.append(dict(case=case, code=ex.code, msg=str(ex), seconds=time.time() - t0))
    if cfg.get("then_good"):                                          # the handle is usable afterwards
        for e, P0 in zip(es, P):
            e.upload_cloud(P0)
        r = eng.run_group(es, **kw) if shards > 1 else es[0].run(**kw)
        out[-1]["good_after"] = [r["n_stages"], bool(np.isfinite(r["logmdd"])), r["n_segments"]]
print("RESULT " + json.dumps(out))
'''


def _run(cfg, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", _W % dict(root=ROOT, cfg=json.dumps(cfg))], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


ENGINES = [("segments", {}, 1), ("launches", {"SMCMI_ENGINE3": "0"}, 1), ("engine1", {"SMCMI_ENGINE": "1"}, 1), ("two_shards", {}, 2)]


@pytest.mark.parametrize("name,env,shards", ENGINES, ids=[e[0] for e in ENGINES])
def test_nan_ess_guard_on_every_engine(name, env, shards):
    cfg = dict(n=20480, d=10, shards=shards, cases=["nan_loglh", "all_minus_inf"], kw=dict(use_fixed_schedule=True, n_phi=40), then_good=True)
    res = _run(cfg, env)
    for r in res:
        assert r["code"] == -3, r                                                    # SMCMI_ERR_NAN_ESS
        assert "No particles have non-zero weight." in r["msg"] and "ESS is NaN" in r["msg"], r
        assert r["seconds"] < 5.0, r
        assert r["good_after"][0] == 40 and r["good_after"][1], r                    # the next run on the same handle is fine
    assert "NaN log-likelihoods" in res[0]["msg"], res[0]                            # helpers.jl:281-283
    assert "returning a NaN" in res[0]["msg"] and "returning a NaN" in res[1]["msg"], res     # helpers.jl:287-292
    if name == "segments":
        assert res[0]["good_after"][2] >= 1                                          # ... and still on engine 3


def test_nan_objective_on_an_adaptive_schedule_is_the_solvers_error():
    """`fzero` on a NaN objective (src/helpers.jl:49): the reference throws from the root finder before the NaN guard is reached."""
    cfg = dict(n=20480, d=10, cases=["nan_loglh"], kw=dict(use_fixed_schedule=False, tempering_target=0.95))
    for env in ({}, {"SMCMI_ENGINE": "1"}):
        r = _run(cfg, env)[0]
        assert r["code"] in (-6, -3), r                                              # SMCMI_ERR_BRACKET (or the guard, if a stage got that far)


@pytest.mark.parametrize("name,env,shards", ENGINES, ids=[e[0] for e in ENGINES])
def test_posdef_exception_aborts_the_run(name, env, shards):
    cfg = dict(n=20480, d=10, shards=shards, cases=["degenerate"], kw=dict(use_fixed_schedule=True, n_phi=40), then_good=True, warm=True)
    r = _run(cfg, env)[0]
    assert r["code"] == -4 and "PosDefException" in r["msg"], r                      # SMCMI_ERR_POSDEF
    assert r["seconds"] < 0.15 + (2.0 if name != "segments" else 0.0), r             # segments: no 200 ms hand-over time-out on the way out
    assert r["good_after"][0] == 40 and r["good_after"][1], r
    if name == "segments":
        assert r["good_after"][2] >= 1
