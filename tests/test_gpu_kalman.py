"""Config 5: linear-Gaussian state-space model with a per-particle Kalman-filter likelihood on the device,
13 parameters, generalized tempering from old (40 periods) to new (80 periods) data.  No reference source exists for this
likelihood (parity unpinned): the device filter is checked against the oracle's, the oracle's against numpy.linalg
(tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from tests import models
from tests.test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu


def _pars(S):
    pars = [S.parameter("rho%d" % k, 0.0, (-0.95, 0.95), prior=S.Uniform(-0.95, 0.95)) for k in range(8)]
    pars += [S.parameter("sig%d" % k, 0.5, (1e-3, 2.0), prior=S.Uniform(0.0, 2.0)) for k in range(4)]
    pars += [S.parameter("mu", 0.0, (-1e5, 1e5), prior=S.Normal(0.0, 5.0))]
    return pars


def test_kalman_device_loglik_vs_oracle():
    from oracle import oracle as orc

    sp = models.kalman_spec()
    eng = make_engine(sp, 2048, seed=4)
    eng.init_from_prior()
    P = eng.download_cloud()
    m = models.oracle_model(sp)
    ll = np.array([orc.loglik(m.lik, P[i, :13]) for i in range(256)])
    np.testing.assert_allclose(P[:256, 13], ll, rtol=1e-10, atol=1e-8)
    Q = orc.initial_draw(m, 2048, seed=4)
    np.testing.assert_allclose(P[:, :13], Q[:, :13], rtol=1e-12, atol=1e-14)      # same Philox prior draws
    np.testing.assert_allclose(P[:, 13], Q[:, 13], rtol=1e-10, atol=1e-8)
    eng.close()


def test_kalman_fixed_schedule_run_vs_oracle():
    from oracle import oracle as orc

    sp = models.kalman_spec()
    m = models.oracle_model(sp)
    n = 2000
    eng = make_engine(sp, n, seed=9, max_stages=64)
    eng.init_from_prior()
    r = eng.run(n_phi=50, use_fixed_schedule=True, n_blocks=2, n_mh_steps=1, alpha=0.9)
    rec = eng.stage_records(r["n_stages"])
    ro = orc.smc_run(m, orc.initial_draw(m, n, seed=9), n_phi=50, n_blocks=2, n_mh_steps=1, alpha=0.9, seed=9, n_threads=8)
    assert r["n_stages"] == ro["n_stages"] == 50
    # (the tolerances of every other run-level comparison, tests/test_gpu_parity.py _compare_runs: the device filter agrees with the
    # oracle's to ~1e-12 per likelihood and no MH decision flips at this size)
    np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-9)
    np.testing.assert_allclose(rec["accept_hist"], ro["accept_hist"], atol=3.0 / n + 1e-12)
    assert r["logmdd"] == pytest.approx(ro["logmdd"], abs=1e-8)
    assert r["resamples"] == ro["resamples"]
    eng.close()


def test_config5_generalized_tempering_old_to_new_data():
    """First estimation on 40 periods, then a tempered update to 80 periods starting from the old cloud (pw = 0, same
    n_parts: smc_main.jl:249-260), adaptive schedule; posterior means move towards the data-generating parameters."""
    import smc_jl_amd as S
    from oracle import oracle as orc

    y = models.kalman_data(80)
    C, R, Z = models.kalman_structure()
    lik = S.LGSSKalman(C, R, Z, models.KALMAN_KAPPA)
    old = np.ascontiguousarray(y[:, :40])
    kw = dict(n_parts=4000, n_phi=60, n_blocks=3, n_mh_steps=1, alpha=0.9, verbose="none", seed=17)
    c_old, _, _ = S.smc(lik, _pars(S), old, use_fixed_schedule=True, **kw)
    c_new, w, W = S.smc(lik, _pars(S), y, old_data=old, old_cloud=c_old, use_fixed_schedule=False, tempering_target=0.9, **kw)
    assert c_new.tempering_schedule[-1] == 1.0 and c_new.ESS[0] == c_old.ESS[-1]
    assert np.all(np.isfinite(c_new.particles))
    # oracle on the same old cloud
    sp = models.kalman_spec(T=80, old_T=40)
    m = models.oracle_model(sp)
    P0, ess0 = orc.tempered_update_cloud(m, c_old.particles, c_old.ESS[-1], 4000, seed=17)
    ro = orc.smc_run(m, P0, n_phi=60, n_blocks=3, n_mh_steps=1, alpha=0.9, use_fixed_schedule=False, tempering_target=0.9, seed=17,
                     initial_ess=ess0, n_threads=8)
    assert c_new.stage_index == ro["n_stages"]
    np.testing.assert_allclose(c_new.tempering_schedule, ro["schedule"], rtol=1e-9)
    np.testing.assert_allclose(c_new.ESS, ro["ess"], rtol=1e-9)
    assert c_new.logmdd == pytest.approx(ro["logmdd"], abs=1e-8)
    mu = S.weighted_mean(c_new)
    assert abs(mu[12] - 1.0) < 0.5 and abs(mu[0] - 0.9) < 0.3          # measurement mean and the most persistent root


def test_config5_sharded_group_matches_single():
    from smc_jl_amd import Engine, run_group

    sp = models.kalman_spec(T=80, old_T=40)
    n = 16384
    e1 = make_engine(sp, n, seed=5, max_stages=400)
    e1.init_from_prior()
    P0 = e1.download_cloud()
    r1 = e1.run(n_phi=40, use_fixed_schedule=False, tempering_target=0.9, n_blocks=2)
    P1 = e1.download_cloud()
    e1.close()
    shards = []
    for k in range(2):
        e = Engine(n, 13, seed=5, n_local=n // 2, gid0=k * (n // 2), max_stages=400)
        e.set_model(sp)
        e.upload_cloud(P0[k * (n // 2):(k + 1) * (n // 2)])
        shards.append(e)
    r2 = run_group(shards, n_phi=40, use_fixed_schedule=False, tempering_target=0.9, n_blocks=2)
    P2 = np.vstack([e.download_cloud() for e in shards])
    for e in shards:
        e.close()
    assert r1["n_stages"] == r2["n_stages"] and r1["resamples"] == r2["resamples"]
    assert r2["logmdd"] == pytest.approx(r1["logmdd"], abs=1e-8)
    same = np.all(np.abs(P1 - P2) <= 1e-8 * (1 + np.abs(P1)), axis=1)
    assert same.mean() > 0.99


_PREFIX_WORKER = r'''
import sys, json, hashlib
sys.path.insert(0, %(root)r)
import numpy as np
from smc_jl_amd import Engine
from tests import models
sp = models.kalman_spec(T=80, old_T=40)
if %(shift)d:                       # an old vintage that is NOT a prefix of the data: two filter passes whatever the switch says
    y = models.kalman_data(80)
    sp["old_lik"] = (sp["old_lik"][0], sp["old_lik"][1], np.ascontiguousarray(y[:, 1:41]), sp["old_lik"][3])
e = Engine(4000, 13, seed=17, max_stages=400, store_history=False)
e.set_model(sp); e.init_from_prior()
r = e.run(n_phi=60, use_fixed_schedule=False, tempering_target=0.9, n_blocks=3, alpha=0.9)
P = e.download_cloud()
print("RESULT " + json.dumps(dict(n_stages=r["n_stages"], logmdd=float(r["logmdd"]).hex(), cloud=hashlib.sha256(np.ascontiguousarray(P).tobytes()).hexdigest())))
'''


@pytest.mark.parametrize("shift", [0, 1])
def test_old_data_prefix_takes_one_filter_pass_with_identical_bits(shift):
    """Tempered update whose old vintage is the first 40 of the 80 periods: the mutation takes both log-likelihoods from one pass
    over the data (csrc/model.hpp kalman_lgss2) - same stages, log-MDD and cloud, bit for bit, as two separate passes
    (SMCMI_NO_LIK_PREFIX=1).  shift = 1: the old vintage is not a prefix; the switch must make no difference."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for off in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", _PREFIX_WORKER % dict(root=root, shift=shift)], env=dict(os.environ, SMCMI_NO_LIK_PREFIX=off),
                           capture_output=True, text=True, timeout=600, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        out.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    assert out[0] == out[1]
    assert out[0]["n_stages"] > 5


_LANES_WORKER = r'''
import sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from smc_jl_amd import Engine
from tests import models
old = Engine(%(n)d, 13, seed=17, max_stages=400, store_history=False)
old.set_model(models.kalman_spec(T=40)); old.init_from_prior()
r0 = old.run(n_phi=60, use_fixed_schedule=False, tempering_target=0.9, n_blocks=%(nb)d, alpha=0.9)
ess0 = float(old.stage_records(r0["n_stages"])["ess"][-1])
old.set_model(models.kalman_spec(T=80, old_T=40))          # tempered update from the old posterior (smc_main.jl:249-260)
old.initialize_likelihoods()
r = old.run(n_phi=60, use_fixed_schedule=False, tempering_target=0.9, n_blocks=%(nb)d, alpha=0.9, initial_ess=ess0)
np.save(%(out)r, old.download_cloud())
print("RESULT " + json.dumps(dict(n_stages=r["n_stages"], resamples=r["resamples"], logmdd=r["logmdd"], old_stages=r0["n_stages"])))
'''


@pytest.mark.parametrize("n,nb", [(4000, 1), (12500, 3)])
def test_four_lanes_per_particle_match_one_thread_per_particle(n, nb, tmp_path):
    """The lane-split Kalman mutation (csrc/model.hpp kalman_lgss_quad: four lanes per particle, the default up to 32 768 particles per
    handle) against one thread per particle (SMCMI_KALMAN_LANES=1) on a whole estimation + tempered update: the filters differ in
    summation order only (1e-13), so stage counts agree and the clouds to 1e-9 (a flipped MH decision would show as an O(1) row
    difference; up to 0.1 % of the rows may)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res, clouds = [], []
    for lanes in ("4", "1"):
        out = str(tmp_path / ("cloud_%s.npy" % lanes))
        p = subprocess.run([sys.executable, "-c", _LANES_WORKER % dict(root=root, n=n, nb=nb, out=out)], env=dict(os.environ, SMCMI_KALMAN_LANES=lanes),
                           capture_output=True, text=True, timeout=900, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]))
        clouds.append(np.load(out))
    assert res[0]["n_stages"] == res[1]["n_stages"] and res[0]["resamples"] == res[1]["resamples"] and res[0]["old_stages"] == res[1]["old_stages"]
    assert res[0]["n_stages"] > 5
    assert res[0]["logmdd"] == pytest.approx(res[1]["logmdd"], abs=1e-8)
    same = np.all(np.abs(clouds[0] - clouds[1]) <= 1e-9 * (1 + np.abs(clouds[1])), axis=1)
    assert same.mean() > 0.999


@pytest.mark.parametrize("n", [4000, 40000])
def test_convergent_filters_match_the_one_thread_filter(n):
    """The two filters whose structure values travel through DPP operands against `kalman_lgss2` on the same parameter vectors:
    `init_from_prior` scores the draws with kalman_lgss2, `initialize_likelihoods` scores them again with the lane-split filter
    (n <= 32 768: four lanes per particle) or with kalman_lgss_wave (larger clouds).  Neither n is a multiple of the block size:
    the last block has lanes without a particle (the case a DPP source lane must not be lost in).  Then parameter vectors whose
    filter explodes: finite values or -Inf, never NaN (a failed filter's state must not reach log())."""
    sp = models.kalman_spec(T=80)
    e = make_engine(sp, n, seed=17)
    e.init_from_prior()
    P0 = e.download_cloud()
    e.initialize_likelihoods()
    P1 = e.download_cloud()
    np.testing.assert_allclose(P1[:, 13], P0[:, 13], rtol=1e-10, atol=1e-9)
    np.testing.assert_array_equal(P1[:, 15], P0[:, 13])                  # old_loglh <- loglh
    np.testing.assert_allclose(P1[:, 14], P0[:, 14], rtol=1e-12, atol=1e-12)
    Q = P0.copy()
    rng = np.random.default_rng(3)
    Q[:, :8] = rng.uniform(-0.95, 0.95, size=(n, 8))
    Q[: n // 2, :8] = 0.94                                              # spectral radius of Tm above one
    e.upload_cloud(Q)
    e.initialize_likelihoods()
    P2 = e.download_cloud()
    assert not np.isnan(P2[:, 13]).any()
    assert np.isfinite(P2[n // 2:, 13]).mean() > 0.5
    m = models.oracle_model(sp)
    from oracle import oracle as orc

    rows = np.r_[0:8, n // 2:n // 2 + 8]
    ref = np.array([orc.loglik(m.lik, Q[i, :13]) for i in rows])
    got = P2[rows, 13]
    both = np.isfinite(ref) & np.isfinite(got)
    np.testing.assert_allclose(got[both], ref[both], rtol=1e-9, atol=1e-7)
    assert np.array_equal(np.isfinite(ref), np.isfinite(got))
    e.close()


def test_config5_tempered_update_at_full_size_against_the_oracle():
    """BASELINE config 5 as `bench.py --workload kalman` runs it - estimation on the old vintage (40 periods, from the prior), then the
    generalized-tempering update to 80 periods from that cloud (src/smc_main.jl:244-333, prior weight 0, same n_parts) - at its full
    N = 50 000 on one handle AND as 4 shards of 12 500 (the four-lane filter), against the oracle's tempered_update_cloud + smc_run on
    the same old cloud and Philox seed, with the tolerances every other run-level comparison uses (tests/test_gpu_parity.py
    _compare_runs): stage and resample counts equal, phi and ESS paths to 1e-9, log-MDD to 1e-8, at most three flipped MH decisions per
    stage in the acceptance rates."""
    import os

    from oracle import oracle as orc
    from smc_jl_amd import Engine, run_group

    n, seed = 50_000, 1
    kw = dict(n_phi=100, use_fixed_schedule=False, tempering_target=0.95, n_blocks=1, n_mh_steps=1, alpha=0.9)
    sp_old, sp = models.kalman_spec(T=40), models.kalman_spec(T=80, old_T=40)
    e = Engine(n, 13, seed=seed, max_stages=600, store_history=False)
    e.set_model(sp_old)
    e.init_from_prior()
    r_old = e.run(**kw)
    P_old = e.download_cloud()
    ess_old = float(e.stage_records(r_old["n_stages"])["ess"][-1])
    # ---- the oracle's update from the same old-vintage cloud
    m = models.oracle_model(sp)
    P_cpu, ess0 = orc.tempered_update_cloud(m, P_old, ess_old, n, seed=seed)
    assert ess0 == ess_old
    ro = orc.smc_run(m, P_cpu, seed=seed, n_threads=os.cpu_count(), history=False, max_stages=600, initial_ess=ess0, **kw)
    assert ro["n_stages"] > 20 and ro["resamples"] >= 1

    def check(r, rec, what):
        assert r["n_stages"] == ro["n_stages"], what
        assert r["resamples"] == ro["resamples"], what
        np.testing.assert_allclose(rec["schedule"], ro["schedule"], rtol=1e-9, err_msg=what)
        np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-9, err_msg=what)
        np.testing.assert_array_equal(rec["resampled"], ro["resampled"], err_msg=what)
        np.testing.assert_allclose(rec["c_hist"], ro["c_hist"], rtol=1e-9, err_msg=what)
        np.testing.assert_allclose(rec["accept_hist"], ro["accept_hist"], atol=3.0 / n + 1e-12, err_msg=what)
        assert abs(r["logmdd"] - ro["logmdd"]) <= 1e-8, (what, r["logmdd"], ro["logmdd"])

    # ---- one handle (the one-thread filter)
    e.set_model(sp)
    e.upload_cloud(P_old)
    e.initialize_likelihoods()
    P_new0 = e.download_cloud()
    np.testing.assert_allclose(P_new0[:, 13], P_cpu[:, 13], rtol=1e-10, atol=1e-8)          # new-vintage log-likelihoods of the old cloud
    np.testing.assert_allclose(P_new0[:, 15], P_cpu[:, 15], rtol=1e-10, atol=1e-8)          # old-vintage ones
    r1 = e.run(initial_ess=ess_old, **kw)
    check(r1, e.stage_records(r1["n_stages"]), "one handle")
    P1 = e.download_cloud()
    e.close()
    mu_g = (P1[:, :13] * P1[:, -1:]).sum(0) / P1[:, -1].sum()
    Pc = ro["particles"]
    mu_c = (Pc[:, :13] * Pc[:, -1:]).sum(0) / Pc[:, -1].sum()
    np.testing.assert_allclose(mu_g, mu_c, atol=1e-6)
    # ---- 4 shards of 12 500 (config 5's stated machine: the four-lane filter, sharded two-launch stage)
    shards = []
    for k in range(4):
        s = Engine(n, 13, seed=seed, n_local=n // 4, gid0=k * (n // 4), max_stages=600, store_history=False)
        s.set_model(sp)
        s.upload_cloud(P_old[k * (n // 4):(k + 1) * (n // 4)])
        s.initialize_likelihoods()
        shards.append(s)
    r4 = run_group(shards, initial_ess=ess_old, **kw)
    check(r4, shards[0].stage_records(r4["n_stages"]), "4 shards of 12 500")
    for s in shards:
        s.close()
