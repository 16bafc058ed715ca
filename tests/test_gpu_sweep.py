"""Randomised device-vs-oracle sweep (fixed seed): dimensions 1..13 (register kernels, fused correction+moments, LDS variants),
1-3 random blocks, 1-2 MH steps, mixture weights, fixed / adaptive schedules, both resamplers, two ESS thresholds.  Every run must
reproduce the oracle's stage count, resample count, ESS path and log-MDD on the same Philox streams."""
import numpy as np
import pytest

from tests import models

pytestmark = pytest.mark.gpu


def test_random_configurations_match_oracle():
    from smc_jl_amd import Engine
    from oracle import oracle as orc

    rs = np.random.RandomState(20260928)
    seen_d = set()
    for trial in range(16):
        d = int(rs.randint(1, 14))
        seen_d.add(d)
        spec = models.gauss_spec(d=d, sigma=float(rs.uniform(0.2, 0.6)))
        nb = int(rs.randint(1, min(d, 3) + 1))
        while ((d + nb - 1) // nb) * (nb - 1) >= d:
            nb -= 1
        kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])),
                  use_fixed_schedule=bool(rs.randint(0, 2)), n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])),
                  resampling_method=str(rs.choice(["systematic", "multinomial"])), threshold_ratio=float(rs.choice([0.5, 0.8])))
        n, seed = int(rs.choice([2048, 4096, 6000])), int(rs.randint(1, 1000))
        e = Engine(n, d, seed=seed, max_stages=1500)
        e.set_model(spec)
        e.init_from_prior()
        P0 = e.download_cloud()
        r = e.run(**kw)
        rec = e.stage_records(r["n_stages"])
        e.close()
        ro = orc.smc_run(models.oracle_model(spec), P0, seed=seed, n_threads=8, max_stages=1500, **kw)
        tag = "trial %d: d=%d n=%d %r" % (trial, d, n, kw)
        assert r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"], tag
        assert r["logmdd"] == pytest.approx(ro["logmdd"], abs=1e-8), tag
        np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-8, err_msg=tag)
        np.testing.assert_allclose(rec["schedule"], ro["schedule"], rtol=1e-9, err_msg=tag)
    assert len(seen_d) >= 8


@pytest.mark.parametrize("name,env,d_max", [
    ("engine 1 (the stage of large single-handle clouds, n_para > 16 and host closures) on small clouds", {"SMCMI_ENGINE": "1"}, 13),
    ("engine 2's large-shard stage (k2b_mutate, helper blocks, self-mailbox) on small clouds", {"SMCMI_ENGINE": "2", "SMCMI_E2_REDUCED": "1"}, 10),
    ("engine 2's launches without segments", {"SMCMI_ENGINE3": "0"}, 10),
])
def test_random_configurations_match_oracle_on_every_stage_engine(name, env, d_max):
    """The sweep above through the stage engines the defaults do not pick at these sizes (the switches are read once per process: a worker
    process per engine) - every (engine x schedule x alpha x resampler) cell is checked against the ORACLE inside the suite, not only
    against another engine (path x evidence matrix)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "sweep_worker.py"), "20260930", "16", str(d_max)], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=1200, cwd=root)
    assert p.returncode == 0 and "DONE 16" in p.stdout, (name, p.stdout[-1500:], p.stderr[-1500:])
    rows = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 16
    seen = set()
    for r in rows:
        tag = "%s: %r" % (name, r)
        assert r["stages"][0] == r["stages"][1] and r["resamples"][0] == r["resamples"][1], tag
        assert r["logmdd_err"] <= 1e-8 and r["ess_relerr"] <= 1e-8 and r["phi_relerr"] <= 1e-9, tag
        seen.add((r["kw"]["use_fixed_schedule"], r["kw"]["alpha"] == 1.0, r["kw"]["resampling_method"]))
    assert len(seen) >= 6, seen                                 # the sweep visited most (schedule, alpha = 1?, resampler) cells


@pytest.mark.parametrize("n", [33, 65, 1023, 100003])
def test_ragged_cloud_sizes_match_oracle(n):
    """Cloud sizes that are no multiple of a wavefront, a block or a chunk (and one below a single wavefront): both schedules,
    two parameter blocks."""
    from smc_jl_amd import Engine
    from oracle import oracle as orc

    spec = models.regression_spec()
    for fixed in (True, False):
        e = Engine(n, 2, seed=9, max_stages=600)
        e.set_model(spec)
        e.init_from_prior()
        P0 = e.download_cloud()
        kw = dict(use_fixed_schedule=fixed, n_phi=40, tempering_target=0.9, n_blocks=2)
        r = e.run(**kw)
        rec = e.stage_records(r["n_stages"])
        P = e.download_cloud()
        e.close()
        ro = orc.smc_run(models.oracle_model(spec), P0, seed=9, n_threads=4, max_stages=600, **kw)
        assert r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"]
        assert r["logmdd"] == pytest.approx(ro["logmdd"], abs=1e-9)
        np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-9)
        same = np.all(np.abs(P - ro["particles"]) <= 1e-8 * (1.0 + np.abs(ro["particles"])), axis=1)
        assert same.mean() > 0.999
