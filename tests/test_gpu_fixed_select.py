"""Fixed schedules on the large-cloud stage (engine 1, csrc/smcmi.hip smcmi_run): every stage is enqueued WITHOUT the selection
kernels (begin, correction + moments, prepare, mutation) while the host stays a bounded number of stages ahead of the device through
host-mapped progress words; a stage that resamples after all stalls and is resumed through the full path without a host sync.
Reference loop: src/smc_main.jl:377-508 (the selection step :435-447 is what the stage leaves out when ESS >= threshold).

Every way of running - the seven-launch stage (SMCMI_FIXED_NO_SELECT=0), the new stage with the host 1 / 2 / 6 stages ahead - must give
the same stages, the same resample stages and the same numbers to rounding (the fused correction + moments pass sums in another
order), and must follow the CPU oracle like the seven-launch stage does."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
from tests import models
from tests.test_gpu_parity import make_engine
cfg = json.loads(sys.argv[1])
spec = models.capm_spec() if cfg["model"] == "capm" else models.gauss_spec(d=cfg["d"])
eng = make_engine(spec, cfg["n"], seed=cfg["seed"], max_stages=400)
eng.init_from_prior()
kw = dict(use_fixed_schedule=True, n_phi=cfg["n_phi"], n_mh_steps=cfg["mh"], n_blocks=cfg["blocks"], resampling_method=cfg["resampler"])
if cfg.get("pause"):
    r0 = eng.run(stop_after_stage=cfg["pause"], **kw)
    assert r0["paused"] == 1
    r = eng.run(continue_run=True, **kw)
    r["select_stalls"] += r0["select_stalls"]
else:
    r = eng.run(**kw)
rec = eng.stage_records(r["n_stages"])
P = eng.download_cloud()
print(json.dumps(dict(n=r["n_stages"], logmdd=r["logmdd"], resamples=r["resamples"], sel=r["select_stalls"], ess=rec["ess"].tolist(),
                      resampled=[int(x) for x in rec["resampled"]], accept=rec["accept_hist"].tolist(), c=rec["c_hist"].tolist(),
                      chk=float(np.sum(P[:, :-5] * np.arange(1, P.shape[1] - 4)[None, :])))))
''' % ROOT


def _run(cfg, env_extra):
    env = dict(os.environ, SMCMI_ENGINE="1", **env_extra)
    res = subprocess.run([sys.executable, "-c", CODE, json.dumps(cfg)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


CASES = [
    dict(model="gauss", d=6, n=20000, seed=11, n_phi=120, mh=1, blocks=1, resampler="systematic"),
    dict(model="gauss", d=10, n=150000, seed=12, n_phi=60, mh=2, blocks=2, resampler="multinomial"),
    dict(model="capm", n=30000, seed=13, n_phi=150, mh=3, blocks=1, resampler="systematic"),
]


@pytest.mark.parametrize("cfg", CASES, ids=["gauss6", "gauss10_2blocks_multinomial", "capm"])
def test_fixed_schedule_without_selection_kernels_equals_the_seven_launch_stage(cfg):
    full = _run(cfg, {"SMCMI_FIXED_NO_SELECT": "0"})
    assert full["sel"] == 0 and full["resamples"] > 0
    for extra in ({},):
        got = _run(cfg, extra)
        # every resample stage was met without its selection kernels and resumed (the first stages of a diffuse prior can resample back to back)
        assert got["sel"] == got["resamples"] == full["resamples"]
        assert got["n"] == full["n"] == cfg["n_phi"]
        assert got["resampled"] == full["resampled"]
        np.testing.assert_allclose(got["ess"], full["ess"], rtol=1e-8)
        np.testing.assert_allclose(got["c"], full["c"], rtol=1e-9)
        np.testing.assert_allclose(got["accept"], full["accept"], atol=3.0 / cfg["n"])
        assert got["logmdd"] == pytest.approx(full["logmdd"], abs=1e-6)
        assert got["chk"] == pytest.approx(full["chk"], rel=1e-5)
    # (the host running ahead by 2 or 6 stages instead of 1 left the same bits - rounds 4 and 5; the switch was retired in round 6)


def test_fixed_schedule_without_selection_kernels_pause_and_continue():
    cfg = dict(CASES[0])
    whole = _run(cfg, {})
    parts = _run(dict(cfg, pause=37), {})
    assert parts["n"] == whole["n"] and parts["resampled"] == whole["resampled"] and parts["sel"] == whole["sel"]
    assert parts["logmdd"] == whole["logmdd"] and parts["ess"] == whole["ess"] and parts["chk"] == whole["chk"]


@pytest.mark.parametrize("case", [0, 1], ids=["gauss6", "gauss10_2blocks_multinomial_n150000"])
def test_fixed_schedule_without_selection_kernels_follows_the_oracle(case):
    """The same run against the CPU restatement (same Philox seed, same initial cloud; the engine under test runs in a child process
    because the library reads its switches once per process)."""
    from oracle import oracle as orc
    from tests import models
    from tests.test_gpu_parity import make_engine

    orc.build()
    cfg = dict(CASES[case])
    spec = models.gauss_spec(d=cfg["d"])
    got = _run(cfg, {})
    eng = make_engine(spec, cfg["n"], seed=cfg["seed"], max_stages=400)
    eng.init_from_prior()
    P0 = eng.download_cloud()
    r = orc.smc_run(models.oracle_model(spec), P0, seed=cfg["seed"], use_fixed_schedule=True, n_phi=cfg["n_phi"], n_mh_steps=cfg["mh"],
                    n_blocks=cfg["blocks"], resampling_method=cfg["resampler"], n_threads=32, history=False)
    assert r["n_stages"] == got["n"]
    assert abs(r["logmdd"] - got["logmdd"]) < 1e-3          # north_star's log-MDD tolerance
    np.testing.assert_allclose(got["ess"], r["ess"], rtol=1e-6)
    assert [int(x) for x in r["resampled"]] == got["resampled"]


def test_random_numbers_partly_drawn_ahead_are_the_same_numbers():
    """4 proposals for each of 150 000 particles do not fit the set-up launch's window: the first k are drawn ahead by its idle CUs
    (csrc/smcmi.hip ensure_zbuf, kernels.hpp rng_ahead_block), the others inside the mutation kernel - pure functions of (seed,
    particle, stage, proposal), so k = 0, 1 and 3 must leave the same bits (reference: the draws of src/mutation.jl:81-101)."""
    cfg = dict(CASES[1])
    runs = [_run(cfg, {"SMCMI_RNG_AHEAD_PART": part}) for part in ("0", "250000", "450000")]
    for r in runs[1:]:
        assert r["n"] == runs[0]["n"] and r["resampled"] == runs[0]["resampled"]
        assert r["logmdd"] == runs[0]["logmdd"] and r["ess"] == runs[0]["ess"] and r["accept"] == runs[0]["accept"] and r["chk"] == runs[0]["chk"]
