"""HIP path (through the C ABI of libsmcmi.so) vs the CPU oracle and the reference's golden vectors.

All tests need a real MI355X (`-m gpu`).  Tolerances: the engine computes in FP64 like the reference;
differences come only from summation order (wavefront/block trees vs sequential) and device libm
(exp/log/sincos within ~1 ulp), so per-function comparisons use rtol 1e-10..1e-12 and discrete
outputs (ancestor indices, accept flags, stage counts) must match exactly.
"""
import numpy as np
import pytest

from tests import models

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle


def make_engine(spec, n, seed=0, **kw):
    from smc_jl_amd import Engine

    e = Engine(n, len(spec["priors"]), seed=seed, **kw)
    e.set_model(spec)
    return e


def dummy_spec(d):
    return dict(priors=[("normal", 0.0, 1.0)] * d, bounds=[(-1e9, 1e9)] * d, fixed=[0] * d,
                lik=("gauss_iso", [1.0], np.zeros((d, 1)), None), old_lik=None)


def random_cloud(rng, n, d, tempered=False):
    P = np.zeros((n, d + 5), order="F")
    P[:, :d] = rng.normal(size=(n, d))
    P[:, d] = -50.0 * rng.random(n) - 3.0          # loglh
    P[:, d + 1] = -rng.random(n)                   # logprior
    P[:, d + 2] = -40.0 * rng.random(n) if tempered else 0.0
    P[:, d + 4] = rng.random(n) * 2.0              # weights
    P[:, d + 4] *= n / P[:, d + 4].sum()
    return P


# ----------------------------------------------------------------------------------------------- ESS / ϕ
def test_ess_at_golden(golden):
    z = golden("ess")
    n = z["loglh"].size
    e = make_engine(dummy_spec(1), n)
    P = np.zeros((n, 6), order="F")
    P[:, 1], P[:, 5] = z["loglh"], z["weights"]
    e.upload_cloud(P)
    ess = e.ess_at([float(z["phi_n"])], float(z["phi_n1"]))[0]
    assert ess == pytest.approx(391.79648393931234, rel=1e-12)


@pytest.mark.parametrize("n,tempered", [(1000, False), (50000, True), (100003, False)])
def test_ess_at_vs_oracle(orc, n, tempered):
    rng = np.random.default_rng(n)
    P = random_cloud(rng, n, 3, tempered)
    e = make_engine(dummy_spec(3), n)
    e.upload_cloud(P)
    phis = np.concatenate([[0.0101], np.linspace(0.011, 0.9, 69)])   # 70 candidates: exercises >1 pass of 32
    got = e.ess_at(phis, 0.01)
    want = [orc.compute_ess(P[:, 3], P[:, 7], ph, 0.01, P[:, 5]) for ph in phis]
    np.testing.assert_allclose(got, want, rtol=1e-11)


def test_solve_phi_golden(golden):
    z = golden("adaptive_phi")
    P = z["particles"]
    n, R = P.shape
    e = make_engine(dummy_spec(R - 5), n)
    e.upload_cloud(P)
    phi_n, rl, j, phi_prop = e.solve_phi(z["schedule"], int(z["j"]), float(z["phi_prop"]), float(z["phi_n1"]),
                                         float(z["target"]), float(z["cloud_ess"][int(z["i"]) - 2]), bool(z["resampled_last"]))
    assert phi_n == pytest.approx(1.212927219006027e-05, rel=1e-9)
    assert j == 3 and phi_prop == float(z["out_phi_prop"]) and rl is False


@pytest.mark.parametrize("case", ["first_stage", "mid_run", "after_resample", "reach_one", "long_scan"])
def test_solve_phi_vs_oracle(orc, case):
    rng = np.random.default_rng(7)
    n, d = 20000, 2
    P = random_cloud(rng, n, d)
    n_phi = 300
    sched = (np.arange(n_phi) / (n_phi - 1.0)) ** 2.1
    if case == "first_stage":
        P[:, d + 4] = 1.0
        args = dict(j=2, phi_prop=0.0, phi_prev=0.0, ess_prev=float(n), rl=False)
    elif case == "mid_run":
        ess0 = orc.compute_ess(P[:, d], P[:, d + 4], 0.2, 0.2)
        args = dict(j=140, phi_prop=sched[138], phi_prev=0.2, ess_prev=ess0, rl=False)
    elif case == "after_resample":
        P[:, d + 4] = 1.0
        args = dict(j=150, phi_prop=sched[148], phi_prev=0.23, ess_prev=0.4 * n, rl=True)
    elif case == "reach_one":
        P[:, d] = -1e-7 * rng.random(n)      # almost flat likelihood: ESS never drops -> ϕ_n = 1
        P[:, d + 4] = 1.0
        args = dict(j=290, phi_prop=sched[288], phi_prev=0.9, ess_prev=float(n), rl=True)
    else:
        P[:, d] = -0.05 * rng.random(n)      # weak likelihood: the scan walks many schedule entries
        P[:, d + 4] = 1.0
        args = dict(j=2, phi_prop=0.0, phi_prev=0.0, ess_prev=float(n), rl=False)
    e = make_engine(dummy_spec(d), n)
    e.upload_cloud(P)
    got = e.solve_phi(sched, args["j"], args["phi_prop"], args["phi_prev"], 0.97, args["ess_prev"], args["rl"])
    want = orc.solve_adaptive_phi(P, args["ess_prev"], sched, args["j"], args["phi_prop"], args["phi_prev"], 0.97, args["rl"])
    assert got[0] == pytest.approx(want[0], rel=1e-9)
    assert got[1] == want[1] and got[2] == want[2] and got[3] == want[3]
    if case == "reach_one":
        assert got[0] == 1.0


# ----------------------------------------------------------------------------------------------- correction
@pytest.mark.parametrize("pw", [0.0, 1.0, 0.3])
def test_correct_vs_oracle(orc, pw):
    rng = np.random.default_rng(11)
    n, d = 30000, 4
    P = random_cloud(rng, n, d, tempered=True)
    e = make_engine(dummy_spec(d), n)
    e.upload_cloud(P)
    st = e.correct(0.35, 0.3, prior_weight=pw, log_prob_old_data=-20.0)
    Q, inc, nw, ess, su = orc.correct(P, 0.35, 0.3, pw, -20.0)
    assert st["ess"] == pytest.approx(ess, rel=1e-11)
    assert st["sum_unnorm"] == pytest.approx(su, rel=1e-12)
    assert st["logz_inc"] == pytest.approx(np.log(su / n), rel=1e-11, abs=1e-13)
    assert st["resample"] == (ess < 0.5 * n)
    out = e.download_cloud()
    np.testing.assert_allclose(out[:, d + 4], nw, rtol=1e-12)
    np.testing.assert_array_equal(out[:, :d + 4], P[:, :d + 4])


def test_replay_99_stages_through_hip(golden):
    """Reference-produced w/W histories (1000 particles, 99 stages): ESS, resample decisions, normalised
    weights and the implied log-MDD -632.7897906595597 reproduced by the HIP correction kernel."""
    z = golden("replay_as1000")
    w, W, ess_ref = z["w"], z["W"], z["ess"]
    N, S = w.shape
    e = make_engine(dummy_spec(1), N)
    logmdd, n_res = 0.0, 0
    for n in range(1, S):
        cloud = np.zeros((N, 6), order="F")
        cloud[:, 1] = np.log(w[:, n])
        cloud[:, 5] = W[:, n - 1]
        e.upload_cloud(cloud)
        st = e.correct(1.0, 0.0)
        assert st["ess"] == pytest.approx(ess_ref[n], rel=1e-11)
        logmdd += st["logz_inc"]
        if st["resample"]:
            n_res += 1
            assert np.all(W[:, n] == 1.0)
        else:
            np.testing.assert_allclose(e.download_cloud()[:, 5], W[:, n], rtol=1e-11)
    assert n_res == 12
    assert logmdd == pytest.approx(-632.7897906595597, abs=1e-8)


# ----------------------------------------------------------------------------------------------- selection
@pytest.mark.parametrize("n", [400, 4097, 100000])
@pytest.mark.parametrize("method", ["systematic", "multinomial"])
def test_resample_vs_oracle(orc, n, method):
    rng = np.random.default_rng(n + 1)
    d = 3
    P = random_cloud(rng, n, d)
    P[rng.integers(0, n, n // 7), d + 4] = 0.0         # zero-weight particles must never be selected
    P[:, d + 4] *= n / P[:, d + 4].sum()
    e = make_engine(dummy_spec(d), n, seed=99)
    for offsets in ("given", "philox"):
        e.upload_cloud(P)
        if offsets == "given":
            off = rng.random(n) if method == "multinomial" else [rng.random()]
            anc = e.resample(method, stage=5, offsets=off)
            want = orc.resample(P[:, d + 4] / n, method, offsets=off)
        else:
            anc = e.resample(method, stage=5)
            want = orc.resample(P[:, d + 4] / n, method, seed=99, stage=5)
        np.testing.assert_array_equal(anc, want)
        assert np.all(P[anc, d + 4] > 0)
        out = e.download_cloud()
        np.testing.assert_array_equal(out[:, :d + 4], P[anc, :d + 4])     # cloud.particles[new_inds, :]
        np.testing.assert_array_equal(out[:, d + 4], 1.0)                   # reset_weights!
        if method == "systematic":
            assert np.all(np.diff(anc) >= 0)


def test_resample_uniform_weights_is_identity():
    n, d = 5000, 2
    rng = np.random.default_rng(5)
    P = random_cloud(rng, n, d)
    P[:, d + 4] = 1.0
    e = make_engine(dummy_spec(d), n)
    for u in (1e-9, 0.5, 0.999999):
        e.upload_cloud(P)
        np.testing.assert_array_equal(e.resample("systematic", offsets=[u]), np.arange(n))


# ----------------------------------------------------------------------------------------------- moments
@pytest.mark.parametrize("n,d", [(777, 2), (60000, 10), (5000, 25)])
def test_moments_vs_oracle(orc, n, d):
    rng = np.random.default_rng(d)
    P = random_cloud(rng, n, d)
    P[:, :d] = P[:, :d] * (1.0 + np.arange(d)) + 100.0 * np.arange(d)     # large means: cancellation test
    P[:, 0] += 0.5 * P[:, d - 1]
    e = make_engine(dummy_spec(d), n)
    e.upload_cloud(P)
    mean, cov = e.moments()
    np.testing.assert_allclose(mean, orc.weighted_mean(P), rtol=1e-12)
    C = orc.weighted_cov(P)
    np.testing.assert_allclose(cov, C, rtol=1e-8, atol=1e-9 * np.abs(C).max())
    # a second call re-centres on the first mean: agreement tightens to rounding
    mean2, cov2 = e.moments()
    np.testing.assert_allclose(cov2, C, rtol=1e-10, atol=1e-12 * np.abs(C).max())
    np.testing.assert_allclose(mean2, mean, rtol=1e-13)


def test_moments_pinned_by_the_reference_mutation_fixture(golden):
    """smcmi_moments on the cloud of test/reference/mutation_inputs.jld2 against the MvNormal(weighted_mean, weighted_cov) the
    reference stored next to it (a-10 pinned by the reference, not only by the oracle)."""
    z = golden("mutation")
    P = np.asfortranarray(z["particles_in"])
    e = make_engine(dummy_spec(9), P.shape[0])
    e.upload_cloud(P)
    e.moments()
    mean, cov = e.moments()                       # (second call: centred on the first mean)
    np.testing.assert_allclose(mean, z["mu"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cov, z["Sigma"], rtol=1e-10, atol=1e-12 * np.abs(z["Sigma"]).max())
    e.close()


# ----------------------------------------------------------------------------------------------- mutation
def _mutation_case(orc, spec, n, n_blocks, n_mh, alpha, c, phi, seed, stage, P=None):
    m = models.oracle_model(spec)
    if P is None:
        P = orc.initial_draw(m, n, seed=seed)
    # a plausible proposal distribution: weighted moments of the cloud itself
    mean, cov = orc.weighted_mean(P), orc.weighted_cov(P)
    fi = m.free_inds
    mu_f, S_f = mean[fi], (cov[np.ix_(fi, fi)] + cov[np.ix_(fi, fi)].T) / 2
    bf, ba, bp = orc.generate_blocks(len(fi), n_blocks, fi, seed, stage)
    want = orc.mutate_cloud(m, P, mu_f, S_f, bf, ba, bp, phi, 0.0, c, alpha, n_mh, seed, stage, n_threads=4)
    e = make_engine(spec, n, seed=seed)
    e.upload_cloud(P)
    acc = e.mutate(mu_f, S_f, bp, bf, phi, 0.0, c, alpha, n_mh, stage)
    got = e.download_cloud()
    return P, want, got, acc


@pytest.mark.parametrize("name,n_blocks,n_mh,alpha", [("gauss", 1, 1, 1.0), ("gauss", 3, 2, 0.9), ("linmodel", 1, 1, 1.0),
                                                    ("linmodel", 2, 3, 0.9), ("capm", 1, 3, 1.0), ("regression", 2, 1, 0.8),
                                                    ("linmodel_tempered", 3, 1, 0.9), ("gauss", 3, 2, 1.0), ("linmodel", 2, 2, 1.0),
                                                    ("linmodel_tempered", 2, 1, 1.0), ("gauss12", 2, 1, 0.9), ("gauss12", 1, 2, 1.0),
                                                    ("gauss20", 3, 1, 0.9)])
def test_mutation_vs_oracle(orc, name, n_blocks, n_mh, alpha):
    spec = {"gauss": models.gauss_spec, "gauss12": lambda: models.gauss_spec(d=12), "gauss20": lambda: models.gauss_spec(d=20), "linmodel": models.linmodel_spec, "capm": models.capm_spec,
            "regression": models.regression_spec, "linmodel_tempered": lambda: models.linmodel_spec(T=100, old_T=50)}[name]()
    n = 20000
    phi = 0.002 if name.startswith("linmodel") or name == "capm" else 0.05
    P, want, got, acc = _mutation_case(orc, spec, n, n_blocks, n_mh, alpha, 0.4, phi, seed=123, stage=7)
    d = len(spec["priors"])
    # accept column is discrete (accepted block lengths / n_free): it must agree exactly - the contracted product build measured 0
    # differing decisions in 820 000 against the uncontracted oracle (tests/test_gpu_strict.py counts them; the strict build must have
    # none); one razor-edge u < eta decision is the most this test lets pass
    flips = np.flatnonzero(got[:, d + 3] != want[:, d + 3])
    assert flips.size <= 1, flips.size
    keep = np.setdiff1d(np.arange(n), flips)
    np.testing.assert_allclose(got[keep, :d + 3], want[keep, :d + 3], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(got[:, d + 4], P[:, d + 4])                # weights untouched
    assert acc == pytest.approx(want[:, d + 3].mean(), abs=2e-4)
    assert 0.0 < acc < n_mh + 1e-9
    moved = np.any(got[:, :d] != P[:, :d], axis=1)
    assert moved.mean() > 0.001


def test_mutation_reject_path_golden(orc, golden):
    """test/mutation.jl fixture: all 400 proposals are rejected; the kernel must hand the particles back untouched."""
    z = golden("mutation")
    spec = models.linmodel_spec(T=100, old_T=100)
    spec["old_lik"] = ("linmodel3", [], z["old_data"], golden("linmodel")["X"])
    P = z["particles_in"]
    e = make_engine(spec, 400, seed=42)
    e.upload_cloud(P)
    bf = (z["blocks_free"] - 1).astype(np.int32)
    bp = np.concatenate([[0], np.cumsum(z["block_sizes"])]).astype(np.int32)
    acc = e.mutate(z["mu"], z["Sigma"], bp, bf, float(z["phi_n"]), float(z["phi_n1"]), float(z["c"]), float(z["alpha"]), 1, 2)
    out, ref = e.download_cloud(), z["particles_out"]
    np.testing.assert_array_equal(out[:, :12], ref[:, :12])
    np.testing.assert_array_equal(out[:, 12], 0.0)
    np.testing.assert_array_equal(out[:, 13], ref[:, 13])
    assert acc == 0.0


def test_mutation_not_posdef_is_an_error(orc):
    from smc_jl_amd.host._lib import SMCMIError

    spec = models.gauss_spec(d=3)
    e = make_engine(spec, 100)
    e.init_from_prior()
    S = np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]])      # indefinite
    with pytest.raises(SMCMIError) as ei:
        e.mutate(np.zeros(3), S, [0, 3], [0, 1, 2], 0.1, 0.0, 0.5, 1.0, 1, 2)
    assert ei.value.code == -4


def test_propose_accept_split_equals_fused(orc):
    """Host-callback split (propose -> host likelihood -> accept) reproduces the fused kernel when the host evaluates
    the same likelihood."""
    spec = models.gauss_spec(d=6)
    m = models.oracle_model(spec)
    n, seed, stage, n_blocks, n_mh, alpha, c, phi = 5000, 31, 4, 2, 2, 0.9, 0.5, 0.07
    P0 = orc.initial_draw(m, n, seed=seed)
    mean, cov = orc.weighted_mean(P0), orc.weighted_cov(P0)
    bf, ba, bp = orc.generate_blocks(6, n_blocks, m.free_inds, seed, stage)
    e1 = make_engine(spec, n, seed=seed)
    e1.upload_cloud(P0)
    e1.mutate(mean, cov, bp, bf, phi, 0.0, c, alpha, n_mh, stage)
    fused = e1.download_cloud()
    e2 = make_engine(spec, n, seed=seed)
    e2.upload_cloud(P0)
    for step in range(n_mh):
        for b in range(n_blocks):
            prop, lpr, qd = e2.propose(mean, cov, bp, bf, b, step, c, alpha, stage)
            ll = np.array([orc.loglik(m.lik, th) for th in prop])
            e2.accept(ll, None, phi, b, step, n_blocks, stage, last=(step == n_mh - 1 and b == n_blocks - 1))
    split = e2.download_cloud()
    same = split[:, 9] == fused[:, 9]
    assert same.mean() > 0.9995
    np.testing.assert_allclose(split[same], fused[same], rtol=1e-10, atol=1e-10)


# ----------------------------------------------------------------------------------------------- initial draw
@pytest.mark.parametrize("name", ["gauss", "linmodel", "capm", "regression"])
def test_init_from_prior_vs_oracle(orc, name):
    spec = {"gauss": models.gauss_spec, "linmodel": models.linmodel_spec, "capm": models.capm_spec,
            "regression": models.regression_spec}[name]()
    n = 3000
    e = make_engine(spec, n, seed=17)
    e.init_from_prior()
    got = e.download_cloud()
    want = orc.initial_draw(models.oracle_model(spec), n, seed=17)
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11)
    d = len(spec["priors"])
    assert np.all(np.isfinite(got[:, d])) and np.all(got[:, d + 4] == 1.0)


def test_linmodel_loglik_prior_golden_through_hip(golden):
    """The 400 stored (θ, loglh, logprior) triples of the reference's test-suite, evaluated by the device likelihood/prior:
    a zero-step-size proposal... is not expressible, so use the host-callback propose path: with c tiny the proposal
    equals θ to rounding and its device log-prior must match the stored one."""
    z = golden("linmodel")
    spec = models.linmodel_spec()
    P = z["init_lik"]
    e = make_engine(spec, 400, seed=1)
    e.upload_cloud(P)
    S = np.eye(9)
    prop, lpr, qd = e.propose(np.zeros(9), S, [0, 9], np.arange(9), 0, 0, 1e-150, 1.0, 2)
    np.testing.assert_array_equal(prop, P[:, :9])
    np.testing.assert_allclose(lpr, P[:, 10], rtol=1e-12)


# ----------------------------------------------------------------------------------------------- whole loop
def _compare_runs(orc, spec, n, seed, tol_logmdd, **kw):
    m = models.oracle_model(spec)
    P0 = orc.initial_draw(m, n, seed=seed)
    r = orc.smc_run(m, P0, seed=seed, n_threads=8, history=True, **kw)
    e = make_engine(spec, n, seed=seed, max_stages=max(r["n_stages"] + 50, kw.get("n_phi", 300)), store_history=True)
    e.upload_cloud(P0)
    g = e.run(**kw)
    assert g["n_stages"] == r["n_stages"]
    assert g["resamples"] == r["resamples"]
    rec = e.stage_records(g["n_stages"])
    # (tolerances: the adaptive root is certified to 1e-12 / verified to 1e-10 relative per stage and carries forward; a flipped MH
    # decision moves an acceptance rate by at most 1 / n - at most three flips per stage are let through)
    np.testing.assert_allclose(rec["schedule"], r["schedule"], rtol=1e-9)
    np.testing.assert_allclose(rec["ess"], r["ess"], rtol=1e-9)
    np.testing.assert_array_equal(rec["resampled"], r["resampled"])
    np.testing.assert_allclose(rec["c_hist"], r["c_hist"], rtol=1e-9)
    np.testing.assert_allclose(rec["accept_hist"], r["accept_hist"], atol=3.0 / n + 1e-12)
    assert g["logmdd"] == pytest.approx(r["logmdd"], abs=tol_logmdd)
    return e, g, r


def test_run_regression_config1(orc):
    """BASELINE config 1 through the HIP engine: examples/regression_model, N=1000, fixed schedule, defaults."""
    spec = models.regression_spec()
    e, g, r = _compare_runs(orc, spec, 1000, 1793, 1e-3)
    assert g["n_stages"] == 300
    w, W = e.history(g["n_stages"])
    h = np.sum(np.log(np.sum(w[:, 1:] * W[:, :-1], axis=0) / 1000))
    assert h == pytest.approx(g["logmdd"], abs=1e-8)            # log-MDD formula a-9 from the stored histories
    assert np.all(w[:, 0] == 0) and np.all(W[:, 0] == 1)
    np.testing.assert_allclose(W.sum(axis=0), 1000.0, rtol=1e-10)
    assert g["logmdd"] == pytest.approx(-99.88901084799365, abs=0.5)
    P = e.download_cloud()
    np.testing.assert_allclose(orc.weighted_mean(P), [1.00018685, 0.99936133], atol=0.15)


def test_run_with_fixed_and_free_regime_columns_vs_oracle(orc):
    """The shape a regime-switching parameter vector has once flattened (src/smc_main.jl:207-234): extra columns behind the base
    parameters, some fixed in some regimes - here a device family over 8 columns of which three are fixed (a base parameter and two
    regime columns), with random blocks and a mixture proposal: free-index bookkeeping, moments and blocks over the free subset only,
    against the oracle on the same Philox streams."""
    spec = models.gauss_spec(d=8)
    spec["fixed"] = [0, 0, 1, 0, 0, 1, 0, 1]
    spec["priors"] = [("normal", 0.3 * k, 5.0) if not f else ("normal", -1.0 + 2.0 * k / 7.0, 5.0) for k, f in enumerate(spec["fixed"])]
    e, g, r = _compare_runs(orc, spec, 12000, 31, 1e-6, use_fixed_schedule=False, tempering_target=0.93, n_blocks=2, alpha=0.9)
    P = e.download_cloud()
    for k in (2, 5, 7):
        assert np.all(P[:, k] == P[0, k])                       # fixed columns never move
    np.testing.assert_allclose(orc.weighted_mean(P), orc.weighted_mean(r["particles"]), atol=1e-6)


def test_run_gauss_adaptive_small(orc):
    spec = models.gauss_spec()
    _compare_runs(orc, spec, 20000, 3, 1e-4, use_fixed_schedule=False, tempering_target=0.97)


def test_run_gauss_multinomial_blocks(orc):
    spec = models.gauss_spec(d=6)
    _compare_runs(orc, spec, 8000, 5, 1e-3, n_phi=60, resampling_method="multinomial", n_blocks=2, n_mh_steps=2, alpha=0.9)


def test_run_linmodel_posterior_mean(orc):
    """test/smc.jl:53-57: posterior mean within 0.5 of [1,1,1,2,2,1,3,3,1] on the reference's test model."""
    spec = models.linmodel_spec()
    e = make_engine(spec, 5000, seed=42, max_stages=120, store_history=False)
    e.init_from_prior()
    g = e.run(n_phi=120, lam=2.0, alpha=0.9, resampling_method="multinomial")
    P = e.download_cloud()
    np.testing.assert_allclose(orc.weighted_mean(P), [1, 1, 1, 2, 2, 1, 3, 3, 1], atol=0.5)
    assert g["n_stages"] == 120


def test_run_capm_config4_vs_oracle(orc):
    """BASELINE config 4 at test size: examples/capm_model (literal likelihood, Uniform priors on σ with bounds, 3 MH steps per
    mutation, fixed schedule n_Φ = 300, defaults otherwise)."""
    spec = models.capm_spec()
    e, g, r = _compare_runs(orc, spec, 10000, 1793, 1e-3, n_mh_steps=3)
    assert g["n_stages"] == 300
    P = e.download_cloud()
    assert np.all(P[:, [2, 5, 8]] >= 1e-5)                       # σ_i stay inside their bounds
    assert 0.0 < g["accept"] < 3.0                               # quirk Q2: accept is normalised by n_free only


def test_run_capm_config4_full_size_properties():
    """Config 4 at its full size (N = 200 000): size-independent properties of the loop."""
    spec = models.capm_spec()
    n = 200000
    e = make_engine(spec, n, seed=1793, max_stages=300, store_history=True)
    e.init_from_prior()
    g = e.run(n_mh_steps=3)
    assert g["n_stages"] == 300 and np.isfinite(g["logmdd"])
    rec = e.stage_records(300)
    np.testing.assert_allclose(rec["schedule"], (np.arange(300) / 299.0) ** 2.1, rtol=1e-14)
    assert np.all(rec["ess"] > 0) and np.all(rec["ess"] <= n * (1 + 1e-12))
    assert np.all(rec["ess"][1:][rec["resampled"][1:] == 0] >= 0.5 * n)          # strict threshold (quirk Q9)
    w, W = e.history(300)
    np.testing.assert_allclose(W.sum(axis=0), float(n), rtol=1e-9)
    assert np.all(W[:, rec["resampled"] == 1] == 1.0)
    h = np.sum(np.log(np.sum(w[:, 1:] * W[:, :-1], axis=0) / n))
    assert h == pytest.approx(g["logmdd"], abs=1e-7)
    P = e.download_cloud()
    assert np.all(np.isfinite(P)) and np.all(P[:, 13] == W[:, -1])
    assert g["accept"] == pytest.approx(P[:, 12].mean(), rel=1e-12)


def test_run_graph_equals_direct(orc):
    spec = models.gauss_spec(d=4)
    out = []
    for ug in (0, 1):
        e = make_engine(spec, 10000, seed=9, max_stages=400, store_history=False)
        e.init_from_prior()
        g = e.run(use_fixed_schedule=False, use_graph=ug)
        out.append((g["n_stages"], g["logmdd"], e.download_cloud()))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    np.testing.assert_array_equal(out[0][2], out[1][2])         # bitwise: same kernels, same order


@pytest.mark.parametrize("d,kw", [(12, dict(use_fixed_schedule=False, tempering_target=0.95)),
                                  (20, dict(n_phi=80, n_blocks=3, alpha=0.9)),
                                  (10, dict(n_phi=60, n_blocks=3, n_mh_steps=2))])
def test_run_other_dimensions_and_blockings(orc, d, kw):
    """d = 12 (register moments + generic LDS mutation), d = 20 (LDS-tiled moments + generic mutation), and the α = 1
    natural-order factor with several blocks and MH steps."""
    spec = models.gauss_spec(d=d)
    _compare_runs(orc, spec, 8000, 23, 1e-3, **kw)


def test_run_deterministic(orc):
    spec = models.gauss_spec(d=5)
    res = []
    for _ in range(2):
        e = make_engine(spec, 30000, seed=77, max_stages=600, store_history=False)
        e.init_from_prior()
        g = e.run(use_fixed_schedule=False)
        res.append((g["logmdd"], e.download_cloud()))
    assert res[0][0] == res[1][0]
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_run_config2_logmdd_vs_oracle(orc):
    """BASELINE config 2 (10-dim Gaussian, N = 100k, adaptive ϕ): |log-MDD_gpu - log-MDD_cpu| <= 1e-3 on identical
    seeds, and both near the analytic -25.3775."""
    spec = models.gauss_spec()
    e, g, r = _compare_runs(orc, spec, 100000, 1, 1e-3, use_fixed_schedule=False, tempering_target=0.97)
    assert g["logmdd"] == pytest.approx(models.gauss_logmdd(), abs=0.1)


def test_solver_stall_resume_is_exact():
    """A stage whose adaptive-ϕ search needs more kernel passes than were enqueued stalls the run (DevState.done = 2) and the
    host resumes it with more passes: the search continues from the stored state, so the results agree with a run that had a
    generous list (to rounding: the resumed stage uses the unfused kernels)."""
    spec = models.gauss_spec(d=6)
    out = []
    for P in (1, 2, 12):
        eng = make_engine(spec, 8192, seed=21, max_stages=2000)
        eng.init_from_prior()
        # phi_rtol < 0 asks for the root to adjacent floats: more passes than the one or two enqueued -> stalls
        r = eng.run(use_fixed_schedule=False, tempering_target=0.9, n_phi=100, solver_passes=P, sync_every=4, phi_rtol=-1.0)
        out.append((r, eng.stage_records(r["n_stages"]), eng.download_cloud()))
        eng.close()
    assert out[0][0]["solver_stalls"] > 0 and out[2][0]["solver_stalls"] == 0
    # a resumed stage runs the unfused kernels (different summation order in the block sums), so agreement is to rounding
    for r, rec, P_ in out[:2]:
        assert r["n_stages"] == out[2][0]["n_stages"] and r["resamples"] == out[2][0]["resamples"]
        assert r["logmdd"] == pytest.approx(out[2][0]["logmdd"], abs=1e-9)
        np.testing.assert_allclose(rec["schedule"], out[2][1]["schedule"], rtol=1e-10)
        np.testing.assert_allclose(rec["ess"], out[2][1]["ess"], rtol=1e-9)
        same = np.all(np.abs(P_ - out[2][2]) <= 1e-9 * (1.0 + np.abs(out[2][2])), axis=1)
        assert same.mean() > 0.999


def test_selection_kernels_skipped_when_no_resample_expected(monkeypatch):
    """On an adaptive schedule the host leaves k_post_correct / k_resample_gather out of stages it expects not to resample
    (the moments kernel takes the decision); a wrong expectation stalls the run and the host resumes the stage with the full
    path.  All three ways of running give the same results (to rounding)."""
    import subprocess, sys, json, os
    code = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
from tests import models
from tests.test_gpu_parity import make_engine
spec = models.gauss_spec(d=6)
eng = make_engine(spec, 8192, seed=33, max_stages=2000)
eng.init_from_prior()
r = eng.run(use_fixed_schedule=False, tempering_target=0.9, n_phi=100, sync_every=8)
rec = eng.stage_records(r["n_stages"])
P = eng.download_cloud()
print(json.dumps(dict(n=r["n_stages"], logmdd=r["logmdd"], resamples=r["resamples"], sel=r["select_stalls"] + r["spec_stalls"], ess=rec["ess"].tolist(),
                      chk=float(np.sum(P * np.arange(1, P.shape[1] + 1)[None, :])))))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("0", "1", "2"):          # predicted / always the full list / deliberately wrong prediction
        # (SMCMI_SEG_SELECT=0: a persistent segment leaves at a stage that must resample instead of resampling in place - the stall path this test is about)
        env = dict(os.environ, SMCMI_NO_SELECT_PREDICT=mode, SMCMI_SEG_SELECT="0")
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out[mode] = json.loads(res.stdout.strip().splitlines()[-1])
    # (stalls = stages resumed by the host: a wrongly expected "no resample" stage, or - mode 0, first stages - a stage whose
    # predicted ϕ could not be used)
    assert out["0"]["sel"] <= 3 and out["1"]["sel"] == 0 and out["2"]["sel"] >= out["2"]["resamples"] > 0
    for mode in ("1", "2"):          # the fused and unfused kernels sum in different orders: agreement to rounding
        assert out[mode]["n"] == out["0"]["n"] and out[mode]["resamples"] == out["0"]["resamples"]
        assert out[mode]["logmdd"] == pytest.approx(out["0"]["logmdd"], abs=1e-9)
        np.testing.assert_allclose(out[mode]["ess"], out["0"]["ess"], rtol=1e-9)
        assert out[mode]["chk"] == pytest.approx(out["0"]["chk"], rel=1e-7)


def test_other_prior_families_device_draw_and_mutation(orc):
    """Gamma / Beta / InverseGamma / RootInverseGamma priors (what rand(parameters) draws for a DSGE parameter vector,
    src/initialization.jl:23-63): the initial draw on the DEVICE (Marsaglia-Tsang unit gammas on the Philox contract, VERDICT r3 missing 4)
    against the oracle's restatement of the same rules and against the distributions themselves; the recursion - prior densities
    included - must follow the oracle from the same cloud.  Through `smc()` the same model runs end to end."""
    import smc_jl_amd as S
    from scipy import stats
    from smc_jl_amd import Engine
    from smc_jl_amd.host import api

    pars = [S.parameter("g", 1.0, (1e-8, 1e5), prior=S.Gamma(2.0, 1.0)),
            S.parameter("b", 0.5, (0.0, 1.0), prior=S.Beta(2.0, 2.0)),
            S.parameter("ig", 1.0, (1e-8, 1e5), prior=S.InverseGamma(3.0, 2.0)),
            S.parameter("rig", 0.5, (1e-8, 1e5), prior=S.RootInverseGamma(4.0, 0.5)),
            S.parameter("n", 0.0, prior=S.Normal(0.0, 2.0))]
    data = np.array([1.5, 0.4, 1.2, 0.6, -0.3])
    lik = S.GaussIso(0.3)
    spec = api._spec_from(pars, lik.spec(data), None)
    n, seed = 8192, 5
    e = Engine(n, 5, seed=seed, max_stages=800)
    e.set_model(spec)
    e.init_from_prior()
    P0 = e.download_cloud()
    np.testing.assert_allclose(P0, orc.initial_draw(models.oracle_model(spec), n, seed=seed), rtol=1e-10, atol=1e-12)
    assert np.all(P0[:, 0] > 0) and np.all((P0[:, 1] > 0) & (P0[:, 1] < 1)) and np.all(P0[:, 2] > 0) and np.all(P0[:, 3] > 0)
    assert np.all(np.isfinite(P0[:, 5])) and np.all(P0[:, 9] == 1.0)
    # the samplers against the distributions (shapes below 1 take the boost G(a) = G(a + 1) U^(1/a)): a likelihood-free model
    pr2 = [S.parameter("g", 1.0, (1e-12, 1e9), prior=S.Gamma(0.6, 2.0)), S.parameter("b", 0.5, (0.0, 1.0), prior=S.Beta(0.5, 3.0)),
           S.parameter("ig", 1.0, (1e-12, 1e9), prior=S.InverseGamma(3.0, 2.0)), S.parameter("rig", 0.5, (1e-12, 1e9), prior=S.RootInverseGamma(5.0, 0.7)),
           S.parameter("g2", 1.0, (1e-12, 1e9), prior=S.Gamma(7.5, 0.25))]
    e2 = Engine(40000, 5, seed=11, max_stages=4, store_history=False)
    e2.set_model(api._spec_from(pr2, S.GaussIso(1e6).spec(np.zeros(5)), None))           # (a flat likelihood: every draw is kept)
    e2.init_from_prior()
    Q = e2.download_cloud()
    e2.close()
    refs = [stats.gamma(0.6, scale=2.0), stats.beta(0.5, 3.0), stats.invgamma(3.0, scale=2.0), None, stats.gamma(7.5, scale=0.25)]
    for k, ref in enumerate(refs):
        x = Q[:, k]
        if ref is None:                                   # σ with ν τ² / σ² ~ χ²(ν)
            assert stats.kstest(5.0 * 0.7 ** 2 / x ** 2, stats.chi2(5.0).cdf).pvalue > 1e-4
        else:
            assert stats.kstest(x, ref.cdf).pvalue > 1e-4, (k, stats.kstest(x, ref.cdf))
    m = models.oracle_model(spec)
    for i in (0, 100, n - 1):                                    # host densities == the oracle's restatement
        assert P0[i, 6] == pytest.approx(orc.logprior(m, P0[i, :5]), abs=1e-12)
    assert abs(P0[:, 0].mean() - 2.0) < 0.1 and abs(P0[:, 1].mean() - 0.5) < 0.02 and abs(P0[:, 2].mean() - 1.0) < 0.1
    kw = dict(use_fixed_schedule=False, tempering_target=0.95, n_phi=100, n_blocks=2, alpha=0.9)
    r = e.run(**kw)
    rec = e.stage_records(r["n_stages"])
    P = e.download_cloud()
    e.close()
    ro = orc.smc_run(m, P0, seed=seed, n_threads=8, max_stages=800, **kw)
    assert r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"]
    assert r["logmdd"] == pytest.approx(ro["logmdd"], abs=1e-8)
    np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-8)
    same = np.all(np.abs(P - ro["particles"]) <= 1e-8 * (1.0 + np.abs(ro["particles"])), axis=1)
    assert same.mean() > 0.999
    assert np.all(P[:, 0] > 0) and np.all((P[:, 1] > 0) & (P[:, 1] < 1))
    # end to end through the mirror
    c, w, W = S.smc(lik, pars, data, n_parts=4096, n_phi=60, use_fixed_schedule=False, tempering_target=0.95, seed=2, verbose="none")
    mu = S.weighted_mean(c)
    assert c.tempering_schedule[-1] == 1.0 and np.all(np.isfinite(c.particles))
    np.testing.assert_allclose(mu, data, atol=0.25)              # σ = 0.3 dominates every prior here (the InverseGamma(3, 2) prior pulls its mean ~0.15 down)


def test_exported_function_wrappers_vs_oracle(orc):
    """The reference's other exports (src/SMC.jl:14-17) as device-backed calls: resample, mvnormal_mixture_draw, mutation,
    initial_draw, get_cloud - each against the oracle's restatement on the same Philox streams."""
    import smc_jl_amd as S

    rs = np.random.RandomState(4)
    w = rs.rand(500)
    for method in ("systematic", "multinomial"):
        for n_out in (None, 320):
            got = S.resample(w, n_parts=n_out, method=method, seed=7, stage=3)
            want = orc.resample(w, method=method, seed=7, stage=3, n_parts=n_out)
            np.testing.assert_array_equal(got, want)
    # one mixture draw of particle 5 at (stage 9, t = 1)
    A = rs.randn(4, 4)
    Sig = A @ A.T + 4 * np.eye(4)
    mu, th = rs.randn(4), rs.randn(4)
    for alpha in (1.0, 0.6):
        got = S.mvnormal_mixture_draw(th, mu, Sig, c=0.4, alpha=alpha, seed=3, pid=5, stage=9, t=1)
        want = orc.mixture_draw(th, mu, Sig, 0.4, alpha, 3, 5, 9, 1)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    # mutation of one particle row of the regression model
    spec = models.regression_spec()
    data = spec["lik"][2]
    pars = [S.parameter("a", 0.0, prior=S.Normal(0, 10)), S.parameter("b", 0.0, prior=S.Normal(0, 10))]
    m = models.oracle_model(dict(priors=[("normal", 0.0, 10.0)] * 2, bounds=[(-1e5, 1e5)] * 2, fixed=[0, 0], lik=("linreg", [1.0], data, None), old_lik=None))
    P = orc.initial_draw(m, 8, 11)
    mu2, S2 = orc.weighted_mean(P), orc.weighted_cov(P)
    bf, ba, bp = orc.generate_blocks(2, 1, np.arange(2, dtype=np.int32), 11, 6)
    Q = orc.mutate_cloud(m, P.copy(order="F"), mu2, S2, bf, ba, bp, 0.3, 0.2, 0.5, 0.9, 2, 11, 6)
    row = S.mutation(S.LinReg(1.0), pars, data, P[3], mu2, S2, 2, [list(bf)], [list(ba)], 0.3, 0.2, c=0.5, alpha=0.9, n_mh_steps=2, seed=11, pid=3, stage=6)
    np.testing.assert_allclose(row, Q[3], rtol=1e-11, atol=1e-12)
    # initial_draw fills a Cloud; get_cloud reads one back
    c = S.initial_draw(S.LinReg(1.0), pars, data, S.Cloud(2, 64), seed=11)
    np.testing.assert_allclose(c.particles, orc.initial_draw(m, 64, 11), rtol=1e-11, atol=1e-11)


def test_c_abi_example_matches_the_python_binding(tmp_path):
    """examples/c_abi_config2.c (plain C over include/smcmi.h, its own process, the system HIP runtime) reproduces the run the
    ctypes binding makes with the same seed: same kernels, same Philox streams, same bits."""
    import json
    import os
    import subprocess

    from smc_jl_amd import Engine

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_config2"
    r = subprocess.run(["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_config2.c"),
                        "-L", os.path.join(ROOT, "smc.jl_amd", "csrc"), "-lsmcmi", "-lm", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "smc.jl_amd", "csrc") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([str(exe), "20000", "7"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    c = json.loads(p.stdout.strip().splitlines()[-1])
    e = Engine(20000, 10, seed=7, max_stages=1500, store_history=False)
    e.set_model(models.gauss_spec(10))
    e.init_from_prior()
    g = e.run(use_fixed_schedule=False, tempering_target=0.97, n_phi=300)
    e.close()
    assert c["n_stages"] == g["n_stages"] and c["resamples"] == g["resamples"]
    assert c["logmdd"] == g["logmdd"] and c["phi_last"] == 1.0
    assert abs(c["mean0"] - (-1.0) * 25 / 25.0625) < 0.02          # posterior mean of θ_0: m_0 s_p² / (s_p² + σ²)


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads, `roofline` included (small cloud, no CPU leg)."""
    import json
    import os
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--nparts", "20000", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "particle-stages/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    # (one GPU: neither weak nor strong scaling is being measured - null; the N > 1 lines say "strong")
    assert d["higher_is_better"] is True and d["scaling"] is None and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["value"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # (a cloud this small runs in engine 3's segments: the quoted kernel is k3_segment, per-stage figures beside the per-launch ones)
    # (`bound` names the real limiter - a chain of dependent work between chip-wide hand-overs, not HBM; `frac` stays the contract's figure)
    assert r["bound"] == "latency" and "valu" in r and "k3_segment" in r["kernel"] and r["mean_stage_us"] > 0 and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0 < r["floor_us"] <= r["mean_stage_us"] and 0 < r["cus_occupied"] <= 256 and r["hand_overs_per_stage"] == 2
    assert r["mutation_kernel_outside_segments"]["launches"] == 0          # every mutation of this run ran inside segments
    assert d["segment_state"] == 1 and d["segment_timeouts"] == 0 and d["segment_blocks"] == r["cus_occupied"]


def test_device_proposal_densities_against_the_reference_fixture(orc, golden):
    """SURVEY §8 a-14 on the HIP side: the reference's own compute_proposal_densities fixture (test/helpers.jl:101-127,
    α = 0.9, a 13-entry block; pins quirk Q1 - the diagonal component's density uses the UNSCALED Σ_ii) through the dense mixture code
    the α < 1 mutation kernels run (kernels.hpp mix_expand / mix_densities), which is structurally different from the literal form the
    oracle restates (one inverse factor and two dense sweeps instead of three forward substitutions)."""
    from smc_jl_amd.host.engine import debug_proposal_densities

    z = golden("proposal_densities")
    q0, q1 = debug_proposal_densities(z["para_draw"], z["para_subset"], z["mu"], z["Sigma"], float(z["c"]), float(z["alpha"]))
    assert q0 == pytest.approx(4.714243032395692, rel=1e-13)
    assert q1 == pytest.approx(4.714241545508865, rel=1e-13)
    assert q0 == pytest.approx(float(z["q0"]), rel=1e-13) and q1 == pytest.approx(float(z["q1"]), rel=1e-13)
    # and against the (fixture-pinned) oracle on random blocks of other sizes / mixture weights, incl. moves far out in the tails
    rs = np.random.RandomState(5)
    for d, alpha, c, far in [(1, 0.9, 0.5, 1.0), (3, 0.5, 0.3, 1.0), (10, 0.9, 0.7, 1.0), (16, 0.2, 0.4, 1.0), (7, 1.0, 0.5, 1.0), (5, 0.9, 0.5, 30.0)]:
        A = rs.standard_normal((d, d))
        S = A @ A.T + 0.3 * np.eye(d)
        mu = rs.standard_normal(d)
        x = mu + far * np.linalg.cholesky(S) @ rs.standard_normal(d)
        xn = x + c * np.linalg.cholesky(S) @ rs.standard_normal(d)
        g0, g1 = debug_proposal_densities(xn, x, mu, S, c, alpha)
        w0, w1 = orc.proposal_densities(xn, x, mu, S, c, alpha)
        assert g0 == pytest.approx(w0, rel=1e-11, abs=1e-11) and g1 == pytest.approx(w1, rel=1e-11, abs=1e-11), (d, alpha, g0, w0, g1, w1)
    with pytest.raises(Exception, match="PosDef"):
        debug_proposal_densities(np.zeros(2), np.zeros(2), np.zeros(2), np.array([[1.0, 2.0], [2.0, 1.0]]), 0.5, 0.9)
