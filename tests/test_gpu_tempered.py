"""Tempered update from an old cloud (src/smc_main.jl:244-333): the initial cloud is built on the device
from the old estimation - same-size continuation, bridge resample + prior draws, or a change of n_parts - and the
recursion then runs on old/new likelihoods.  Checked against the oracle's restatement of the same steps."""
import numpy as np
import pytest

from tests import models

pytestmark = pytest.mark.gpu

OLD_T = 60


def _pars(S):
    return [S.parameter("α1", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False),
            S.parameter("β1", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False)]


def _spec(data, old):
    return dict(priors=[("normal", 0.0, 10.0)] * 2, bounds=[(-1e5, 1e5)] * 2, fixed=[0, 0],
                lik=("linreg", [1.0], data, None), old_lik=("linreg", [1.0], old, None))


@pytest.fixture(scope="module")
def old_run():
    import smc_jl_amd as S
    data = models.regression_spec()["lik"][2]
    old = np.ascontiguousarray(data[:OLD_T])
    cloud, _, _ = S.smc(S.LinReg(1.0), _pars(S), old, n_parts=4000, n_phi=60, use_fixed_schedule=True, seed=5, verbose="none")
    return data, old, cloud


def _rows_match(P, Q, tol=1e-9):
    same = np.all(np.abs(P - Q) <= tol * (1.0 + np.abs(Q)), axis=1)
    return same.mean()


@pytest.mark.parametrize("n_parts,pw,method", [(4000, 0.0, "systematic"), (4000, 0.3, "systematic"), (3000, 0.0, "systematic"),
                                               (5000, 0.25, "multinomial"), (4000, 1.0, "systematic")])
def test_tempered_initial_cloud_vs_oracle(old_run, n_parts, pw, method):
    import smc_jl_amd as S
    from smc_jl_amd import Engine
    from oracle import oracle as orc
    import importlib
    api = importlib.import_module(S.smc.__module__)

    data, old, cloud = old_run
    sp = _spec(data, old)
    eng = Engine(n_parts, 2, seed=11, max_stages=4)
    eng.set_parameters(sp["priors"], sp["bounds"], sp["fixed"])
    eng.set_likelihood(*sp["lik"], which=0)
    eng.set_likelihood(*sp["old_lik"], which=1)
    def prior_engine(n_pr):                 # prior draws scored by the old likelihood on the old data (smc_main.jl:288-291)
        pri = Engine(n_pr, 2, seed=11, max_stages=2, store_history=False)
        pri.set_parameters(sp["priors"], sp["bounds"], sp["fixed"])
        pri.set_likelihood(*sp["old_lik"], which=0)
        pri.set_likelihood("none", which=1)
        pri.init_from_prior()
        return pri

    ess0 = api._tempered_update_cloud(eng, cloud, n_parts, pw, method, 11, 0, prior_engine)
    P = eng.download_cloud()
    eng.close()
    m = models.oracle_model(sp)
    Q, ess0_o = orc.tempered_update_cloud(m, cloud.particles, cloud.ESS[-1], n_parts, prior_weight=pw, resampling_method=method,
                                          seed=11)
    assert ess0 == ess0_o
    assert P.shape == Q.shape == (n_parts, 7)
    # rows agree except where a cumulative-weight rounding moved an ancestor by one (scan order differs)
    assert _rows_match(P, Q) > 0.995
    # old_loglh column = old likelihood of the particle, loglh = new likelihood, weights reset/kept as the reference does
    if pw == 0.0 and n_parts == 4000:
        np.testing.assert_array_equal(P[:, 6], cloud.particles[:, 6])
        np.testing.assert_array_equal(P[:, 4], cloud.particles[:, 2])
    else:
        np.testing.assert_array_equal(P[:, 6], 1.0)
    th = P[:, :2]
    e_new = data[:, 0][None, :] - th[:, :1] - th[:, 1:2] * data[:, 1][None, :]
    ll_new = -(len(data) / 2) * np.log(2 * np.pi) - 0.5 * np.sum(e_new ** 2, axis=1)
    np.testing.assert_allclose(P[:, 2], ll_new, rtol=1e-10)
    e_old = old[:, 0][None, :] - th[:, :1] - th[:, 1:2] * old[:, 1][None, :]
    ll_old = -(len(old) / 2) * np.log(2 * np.pi) - 0.5 * np.sum(e_old ** 2, axis=1)
    np.testing.assert_allclose(P[:, 4], ll_old, rtol=1e-10)


@pytest.mark.parametrize("n_parts,pw,fixed", [(4000, 0.0, True), (4000, 0.3, True), (3000, 0.0, False), (4000, 0.5, False)])
def test_tempered_update_run_vs_oracle(old_run, n_parts, pw, fixed):
    import smc_jl_amd as S
    from oracle import oracle as orc

    data, old, cloud = old_run
    kw = dict(n_parts=n_parts, n_phi=40, use_fixed_schedule=fixed, tempering_target=0.9, seed=11, verbose="none",
              tempered_update_prior_weight=pw, log_prob_old_data=-3.0 if pw > 0 else 0.0)
    c, w, W = S.smc(S.LinReg(1.0), _pars(S), data, old_data=old, old_cloud=cloud, **kw)
    sp = _spec(data, old)
    m = models.oracle_model(sp)
    Q, ess0 = orc.tempered_update_cloud(m, cloud.particles, cloud.ESS[-1], n_parts, prior_weight=pw, seed=11)
    r = orc.smc_run(m, Q, n_phi=40, use_fixed_schedule=fixed, tempering_target=0.9, prior_weight=pw,
                    log_prob_old_data=kw["log_prob_old_data"], seed=11, initial_ess=ess0, n_threads=2)
    assert c.stage_index == r["n_stages"]
    assert c.ESS[0] == ess0
    np.testing.assert_allclose(c.tempering_schedule, r["schedule"], rtol=1e-9)
    np.testing.assert_allclose(c.ESS, r["ess"], rtol=1e-8)
    assert c.logmdd == pytest.approx(r["logmdd"], abs=1e-8)
    assert c.tempering_schedule[-1] == 1.0
    # posterior of the full sample
    np.testing.assert_allclose(S.weighted_mean(c), [1.0, 1.0], atol=0.25)
    # first W column rule of a tempered update (smc_main.jl:364-365)
    if pw == 0.0 and n_parts == 4000:
        w0 = cloud.particles[:, 6]
        np.testing.assert_allclose(W[:, 0], w0 * n_parts if w0.sum() <= 1.0 else w0, rtol=1e-14)
    else:
        np.testing.assert_array_equal(W[:, 0], 1.0)


def test_reference_bridge_scenario_linear_model():
    """test/smc.jl:95-140: linear model estimated on the first half of the sample with 1000 particles, then a tempered
    update on the full sample with the default 5000 particles (n_parts differs -> bridge branch), polyalgo resampler,
    α = 0.9, n_Φ = 100; the reference asserts posterior means within 0.5 of the true parameters."""
    import smc_jl_amd as S

    sp = models.linmodel_spec(T=100)
    data, X = sp["lik"][2], sp["lik"][3]
    half = np.ascontiguousarray(data[:, :data.shape[1] // 2])
    pars = []
    for k in range(3):
        pars += [S.parameter("a%d" % k, 0.0, (-1e5, 1e5), prior=S.Normal(0, 1e3)),
                 S.parameter("b%d" % k, 0.0, (-1e5, 1e5), prior=S.Normal(0, 1e3)),
                 S.parameter("s%d" % k, 1.0, (1e-5, 1e5), prior=S.Uniform(0, 1e3))]
    kw = dict(verbose="none", use_fixed_schedule=True, n_phi=100, n_mh_steps=1, resampling_method="polyalgo", target=0.25,
              alpha=0.9, threshold_ratio=0.5, seed=42)
    old_cloud, _, _ = S.smc(S.LinModel3(X), pars, half, n_parts=1000, **kw)
    new_cloud, w, W = S.smc(S.LinModel3(X), pars, data, old_data=half, old_cloud=old_cloud, **kw)
    assert len(new_cloud) == 5000 and new_cloud.stage_index == 100
    assert new_cloud.ESS[0] == 5000.0
    truth = np.array([1., 1., 1., 2., 2., 1., 3., 3., 1.])
    assert np.max(np.abs(S.get_vals(new_cloud).mean(axis=1) - truth)) < 0.5
    assert np.max(np.abs(S.weighted_mean(new_cloud) - truth)) < 0.5
    assert np.all(new_cloud.particles[:, 9 + 2] != 0.0)          # old_loglh column filled by initialize_likelihoods!


def test_bridge_with_large_energies_keeps_the_reference_range():
    """A 40-period extension of the linear model: loglh - old_loglh is ~ -600 for every particle and the first stage's schedule
    walk evaluates ESS up to δ = 1.  The reference normalises the incremental weights before squaring them
    (src/helpers.jl:173-181), so it survives to |δ e| ~ 745; the device shifts the energies by their maximum (kernels.hpp
    stage_shift) instead - unshifted sums of squares used to underflow at ~354 and abort this run with a bracket error."""
    from smc_jl_amd import Engine
    from oracle import oracle as orc

    n, d, seed = 4096, 9, 779
    spec = models.linmodel_spec(T=100, old_T=60)
    e = Engine(n, d, seed=seed, max_stages=400)
    e.set_model(models.linmodel_spec(T=60))
    e.init_from_prior()
    r_old = e.run(n_phi=60, use_fixed_schedule=True, n_mh_steps=2)
    ess_old = float(e.stage_records(r_old["n_stages"])["ess"][-1])
    P_old = e.download_cloud()
    e.close()
    e = Engine(n, d, seed=seed + 1, max_stages=400)
    e.set_model(spec)
    e.upload_cloud(P_old)
    e.initialize_likelihoods()
    P0 = e.download_cloud()
    energy = P0[:, d] - P0[:, d + 2]
    assert np.max(energy) < -360.0                      # beyond what unshifted squares can hold at δ = 1
    kw = dict(n_blocks=3, n_mh_steps=2, alpha=0.5, use_fixed_schedule=False, n_phi=30, tempering_target=0.95,
              resampling_method="multinomial", threshold_ratio=0.8, initial_ess=ess_old)
    r = e.run(**kw)
    rec = e.stage_records(r["n_stages"])
    w, W = e.history(r["n_stages"])
    e.close()
    ro = orc.smc_run(models.oracle_model(spec), P0, seed=seed + 1, n_threads=8, max_stages=400, **kw)
    assert r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"]
    assert r["logmdd"] == pytest.approx(ro["logmdd"], abs=1e-8)
    np.testing.assert_allclose(rec["schedule"], ro["schedule"], rtol=1e-9)
    np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-8)
    # the stored incremental weights are the reference's exp(δ e), not the shifted ones
    np.testing.assert_allclose(w[:, 1], ro["w"][:, 1], rtol=1e-9, atol=0.0)
    np.testing.assert_allclose(W[:, 1], ro["W"][:, 1], rtol=1e-9, atol=1e-300)


def test_intermediate_save_and_continue(tmp_path):
    """save_intermediate / continue_intermediate (src/smc_main.jl:334-361, 499-507): the device loop pauses at every save point,
    the host stores {cloud, w, W, j}, and a new process continues from the file.  Pausing must not change the run; a
    continuation restores i, j, c, ϕ_prop = schedule[j] (resampled_last_period restarts as false, like the reference's)."""
    import smc_jl_amd as S

    data = models.regression_spec()["lik"][2]
    base = dict(n_parts=4000, n_phi=40, tempering_target=0.9, seed=21, verbose="none")
    for fixed in (True, False):
        c0, w0, W0 = S.smc(S.LinReg(1.0), _pars(S), data, use_fixed_schedule=fixed, **base)
        sp = str(tmp_path / ("run_%d.npz" % fixed))
        c1, w1, W1 = S.smc(S.LinReg(1.0), _pars(S), data, use_fixed_schedule=fixed, save_intermediate=True,
                           intermediate_stage_increment=7, savepath=sp, particle_store_path=str(tmp_path / "draws.npy"), **base)
        # pausing only re-enters the stage chain: same stages, same resamples, results to rounding (the first stage after a
        # pause runs the certificate path instead of predict-correct-verify)
        assert c1.stage_index == c0.stage_index and c1.resamples == c0.resamples
        np.testing.assert_allclose(c1.tempering_schedule, c0.tempering_schedule, rtol=1e-9)
        np.testing.assert_allclose(c1.ESS, c0.ESS, rtol=1e-8)
        assert c1.logmdd == pytest.approx(c0.logmdd, abs=1e-8)
        np.testing.assert_allclose(W1, W0, rtol=1e-7, atol=1e-12)
        assert np.load(str(tmp_path / "draws.npy")).shape == (4000, 2)
        saved = sorted(int(p.name.split("_stage=")[1].split(".")[0]) for p in tmp_path.glob("run_%d_stage=*.npz" % fixed))
        assert saved == list(range(7, c0.stage_index, 7))
        # continue from a file whose stage did not resample (the flag the reference drops would matter otherwise)
        rs_flags = np.load(sp)["ESS"]            # final file exists and holds the whole ESS path
        assert rs_flags.size == c0.stage_index
        k = next(s for s in saved if np.load(str(tmp_path / ("run_%d_stage=%d.npz" % (fixed, s))))["resampled"][s - 1] == 0)
        lp = str(tmp_path / ("run_%d_stage=%d.npz" % (fixed, k)))
        c2, w2, W2 = S.smc(S.LinReg(1.0), _pars(S), data, use_fixed_schedule=fixed, continue_intermediate=True, loadpath=lp, **base)
        assert c2.stage_index == c0.stage_index and c2.resamples == c0.resamples
        np.testing.assert_allclose(c2.tempering_schedule, c0.tempering_schedule, rtol=1e-9)
        np.testing.assert_allclose(c2.ESS, c0.ESS, rtol=1e-8)
        assert c2.logmdd == pytest.approx(c0.logmdd, abs=1e-8)
        np.testing.assert_allclose(w2[:, :k], w0[:, :k], rtol=0, atol=0)          # history of the stages done before the save
        np.testing.assert_allclose(S.weighted_mean(c2), S.weighted_mean(c0), rtol=1e-7)


def test_continue_run_error_paths():
    """continue_run needs loop state to continue from; a finished run cannot be continued."""
    from smc_jl_amd import Engine
    from smc_jl_amd.host._lib import SMCMIError

    spec = models.gauss_spec(d=3)
    e = Engine(2048, 3, seed=4, max_stages=400)
    e.set_model(spec)
    e.init_from_prior()
    r = e.run(n_phi=20, use_fixed_schedule=True, stop_after_stage=5)
    assert r["paused"] and r["n_stages"] == 5
    ls = e.get_loop_state()
    assert ls["stage_index"] == 5 and 0.0 < ls["phi_n"] < 1.0 and ls["j"] >= 2
    r = e.run(n_phi=20, use_fixed_schedule=True, continue_run=True)
    assert not r["paused"] and r["n_stages"] == 20
    with pytest.raises(SMCMIError, match="STATE"):
        e.run(n_phi=20, use_fixed_schedule=True, continue_run=True)          # already at phi = 1
    with pytest.raises(SMCMIError, match="ARG"):
        e.set_loop_state(stage_index=0, j=2, phi_n=0.1)
    e.close()


def test_tempered_update_loads_old_cloud_from_loadpath(tmp_path, old_run):
    """cloud_isempty(old_cloud) ? load(loadpath, "cloud") : old_cloud (src/smc_main.jl:245-246); `testing` suppresses the files."""
    import smc_jl_amd as S

    data, old, cloud = old_run
    path = str(tmp_path / "old_cloud.npz")
    S.save_cloud(path, cloud, np.zeros((len(cloud), 1)), np.ones((len(cloud), 1)))
    kw = dict(n_parts=4000, n_phi=40, use_fixed_schedule=True, seed=11, verbose="none", old_data=old)
    a, _, _ = S.smc(S.LinReg(1.0), _pars(S), data, old_cloud=cloud, savepath=str(tmp_path / "a.npz"), testing=True, **kw)
    b, _, _ = S.smc(S.LinReg(1.0), _pars(S), data, loadpath=path, data_vintage="200101", smc_iteration=2, parallel=True, **kw)
    assert not (tmp_path / "a.npz").exists()
    assert a.stage_index == b.stage_index and a.logmdd == b.logmdd
    np.testing.assert_array_equal(a.particles, b.particles)
    with pytest.raises(ValueError):
        S.smc(S.LinReg(1.0), _pars(S), data, regime_switching=True, **kw)
