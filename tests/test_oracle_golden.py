"""Pin the CPU oracle against the reference's own golden fixtures (CPU only, no GPU).

Every value here comes from a file the reference's test-suite loads (see tests/golden/make_fixtures.py
for provenance) or from a closed form.  RNG-free arithmetic must match to ~1e-13; RNG-dependent pieces
are checked through invariants/distributions only (the reference's RNG goldens are MersenneTwister-bound).
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import models


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    import ctypes as C

    out = (C.c_uint32 * 4)()
    L = orc.lib()
    L.orc_philox4x32_10(0, 0, 0, 0, 0, 0, out)
    assert [hex(x) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = 0xFFFFFFFF
    L.orc_philox4x32_10(f, f, f, f, f, f, out)
    assert [hex(x) for x in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    L.orc_philox4x32_10(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0, out)
    assert [hex(x) for x in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_compute_ess_golden(golden):
    z = golden("ess")  # test/helpers.jl:133-175 (the test passes the default old_loglh = zeros)
    ess = orc.compute_ess(z["loglh"], z["weights"], float(z["phi_n"]), float(z["phi_n1"]))
    assert ess == pytest.approx(float(z["ess"]), rel=1e-13)
    assert float(z["ess"]) == pytest.approx(391.79648393931234, rel=1e-15)


def test_solve_adaptive_phi_golden(golden):
    z = golden("adaptive_phi")  # test/helpers.jl:15-53
    i = int(z["i"])
    phi_n, rl, j, phi_prop, n_evals = orc.solve_adaptive_phi(
        z["particles"], float(z["cloud_ess"][i - 2]), z["schedule"], int(z["j"]), float(z["phi_prop"]),
        float(z["phi_n1"]), float(z["target"]), bool(z["resampled_last"]))
    assert phi_n == pytest.approx(float(z["out_phi_n"]), rel=1e-12)
    assert phi_n == pytest.approx(1.212927219006027e-05, rel=1e-12)
    assert j == int(z["out_j"]) == 3
    assert phi_prop == float(z["out_phi_prop"])
    assert rl == bool(z["out_resampled_last"])
    assert n_evals < 80


def test_proposal_densities_golden(golden):
    z = golden("proposal_densities")  # test/helpers.jl:101-127; pins quirk Q1 (alpha = 0.9)
    q0, q1 = orc.proposal_densities(z["para_draw"], z["para_subset"], z["mu"], z["Sigma"], float(z["c"]),
                                    float(z["alpha"]))
    assert q0 == pytest.approx(float(z["q0"]), rel=1e-13)
    assert q1 == pytest.approx(float(z["q1"]), rel=1e-13)
    assert float(z["q0"]) == pytest.approx(4.714243032395692, rel=1e-15)


def test_linmodel_loglik_and_prior_golden(golden):
    z = golden("linmodel")  # test/initialization.jl:29-103 stored (theta, loglh, logprior) triples
    m = models.oracle_model(models.linmodel_spec())
    for key in ("initial_draw", "init_lik"):
        P = z[key]
        for row in P:
            th = row[:9]
            assert orc.loglik(m.lik, th) == pytest.approx(row[9], rel=1e-12)
            assert orc.logprior(m, th) == pytest.approx(row[10], rel=1e-12)
    th = z["one_draw_theta"]
    assert orc.loglik(m.lik, th) == pytest.approx(float(z["one_draw_loglh"][0]), rel=1e-12)
    assert orc.logprior(m, th) == pytest.approx(float(z["one_draw_logprior"][0]), rel=1e-12)
    assert float(z["draw_lik_loglh"][0]) == pytest.approx(-14097.66668904, rel=1e-9)


def test_mutation_reject_path_golden(golden):
    z = golden("mutation")  # test/mutation.jl:22-59: every proposal is rejected in this fixture
    spec = models.linmodel_spec(T=100, old_T=100)
    spec["old_lik"] = ("linmodel3", [], z["old_data"], np.load(models.GOLDEN + "/linmodel.npz")["X"])
    m = models.oracle_model(spec)
    P = z["particles_in"]
    bf = (z["blocks_free"] - 1).astype(np.int32)
    ba = (z["blocks_all"] - 1).astype(np.int32)
    bp = np.concatenate([[0], np.cumsum(z["block_sizes"])]).astype(np.int32)
    out = orc.mutate_cloud(m, P, z["mu"], z["Sigma"], bf, ba, bp, float(z["phi_n"]), float(z["phi_n1"]),
                           float(z["c"]), float(z["alpha"]), 1, seed=42, stage=2)
    ref = z["particles_out"]
    np.testing.assert_array_equal(out[:, :12], ref[:, :12])     # params, loglh, logprior, old_loglh untouched
    np.testing.assert_array_equal(out[:, 12], 0.0)              # accept column
    np.testing.assert_array_equal(ref[:, 12], 0.0)
    np.testing.assert_array_equal(out[:, 13], ref[:, 13])       # weight untouched
    lp = np.array([orc.logprior(m, r[:9]) for r in P])
    np.testing.assert_allclose(lp, P[:, 10], rtol=1e-12)


def test_replay_99_stages_golden(golden):
    """Correction / normalisation / ESS / resample decision / log-MDD bookkeeping against a saved
    1000-particle reference run (w, W histories; src/smc_main.jl:401-446)."""
    z = golden("replay_as1000")
    w, W, ess_ref = z["w"], z["W"], z["ess"]
    N, S = w.shape
    logmdd, n_res = 0.0, 0
    for n in range(1, S):
        cloud = np.zeros((N, 6), order="F")       # d = 1 dummy parameter
        cloud[:, 1] = np.log(w[:, n])              # loglh s.t. exp((1-0)*loglh) = w (1 ulp)
        cloud[:, 5] = W[:, n - 1]
        _, inc, nw, ess, su = orc.correct(cloud, 1.0, 0.0)
        np.testing.assert_allclose(inc, w[:, n], rtol=4e-16)
        assert ess == pytest.approx(ess_ref[n], rel=1e-12)
        logmdd += np.log(su / N)
        resampled = ess < 0.5 * N
        if resampled:
            n_res += 1
            np.testing.assert_array_equal(W[:, n], 1.0)
        else:
            np.testing.assert_allclose(nw, W[:, n], rtol=1e-12)
    assert n_res == int(z["resamples"]) == 12
    assert logmdd == pytest.approx(float(z["logmdd"]), abs=1e-9)
    assert float(z["logmdd"]) == pytest.approx(-632.7897906595597, abs=1e-12)
    assert float(z["accept"]) == pytest.approx(float(np.mean(z["accept_col"])), rel=1e-14)
    np.testing.assert_allclose(z["schedule"], (np.arange(100) / 99.0) ** 2, rtol=1e-14)


# ----------------------------------------------------------------------------- RNG-dependent: invariants
def test_systematic_resample_invariants():
    rng = np.random.default_rng(0)
    w = rng.random(400)
    idx = orc.resample(w, "systematic", seed=42, stage=3)
    assert np.all(np.diff(idx) >= 0) and idx.min() >= 0 and idx.max() < 400
    # uniform weights: identity for any offset
    for u in (1e-9, 0.3, 0.999999):
        np.testing.assert_array_equal(orc.resample(np.ones(257), "systematic", offsets=[u]), np.arange(257))
    # offspring counts within 1 of N*w
    cnt = np.bincount(idx, minlength=400)
    assert np.all(np.abs(cnt - 400 * w / w.sum()) < 1.0 + 1e-9)


def test_multinomial_resample_matches_searchsorted():
    rng = np.random.default_rng(1)
    w = rng.random(300)
    u = rng.random(300)
    idx = orc.resample(w, "multinomial", offsets=u)
    cw = np.cumsum(w / w.sum())
    np.testing.assert_array_equal(idx, np.minimum(np.searchsorted(cw, u, side="right"), 299))


def test_blocks_partition():
    free = np.array([0, 1, 3, 4, 5, 7, 8], dtype=np.int32)
    for nb in (1, 2, 3):
        bf, ba, bp = orc.generate_blocks(7, nb, free, seed=5, stage=9)
        assert sorted(bf.tolist()) == list(range(7))
        np.testing.assert_array_equal(ba, free[bf])
        sizes = np.diff(bp)
        sub = -(-7 // nb)
        assert list(sizes[:-1]) == [sub] * (nb - 1) and sizes[-1] == 7 - sub * (nb - 1)


def test_mixture_draw_distribution(golden):
    z = golden("mvnormal_inputs")  # test/helpers.jl:58-80 inputs; outputs are Julia-RNG bound
    mu, S, c, alpha, th = z["mu"], z["Sigma"], float(z["c"]), float(z["alpha"]), z["para_subset"]
    n = 40000
    X = np.array([orc.mixture_draw(th, mu, S, c, alpha, seed=7, pid=i, stage=2, t=0) for i in range(n)])
    # mixture mean = (alpha + (1-alpha)/2) th + (1-alpha)/2 mu
    m_exp = (alpha + (1 - alpha) / 2) * th + (1 - alpha) / 2 * mu
    se = np.sqrt(np.diag(S) * c * c / n) + np.abs(th - mu) * 0.3 / np.sqrt(n)
    assert np.all(np.abs(X.mean(0) - m_exp) < 6 * se + 1e-12)


def test_update_c_fixed_point():
    assert orc.update_c(0.5, 0.25, 0.25) == pytest.approx(0.5, rel=1e-15)   # quirk Q10: first stage multiplier 1.0


def test_weighted_moments_match_numpy():
    rng = np.random.default_rng(3)
    P = np.asfortranarray(rng.normal(size=(500, 9)))
    P[:, 8] = rng.random(500) * 2
    w = P[:, 8] / P[:, 8].sum()
    X = P[:, :4]
    m = w @ X
    C = (X - m).T @ ((X - m) * w[:, None])
    np.testing.assert_allclose(orc.weighted_mean(P), m, rtol=1e-12)
    np.testing.assert_allclose(orc.weighted_cov(P), C, rtol=1e-11, atol=1e-14)


# ----------------------------------------------------------------------------- end-to-end vs analytic truths
def test_weighted_moments_pinned_by_the_reference_mutation_fixture(golden):
    """a-10 (`weighted_mean`, `weighted_cov`, src/particle.jl:481-483, 526-529) against the reference itself: the MvNormal `d` stored in
    test/reference/mutation_inputs.jld2 is MvNormal(weighted_mean(cloud), weighted_cov(cloud)) of the cloud stored next to it
    (test/mutation.jl:22-34; the cloud was written right after a resample, all weights 1)."""
    z = golden("mutation")
    P = np.asfortranarray(z["particles_in"])
    assert np.all(P[:, -1] == 1.0)
    np.testing.assert_allclose(orc.weighted_mean(P), z["mu"], rtol=1e-12, atol=1e-12)
    C = orc.weighted_cov(P)
    np.testing.assert_allclose(C, z["Sigma"], rtol=1e-12, atol=1e-12 * np.abs(z["Sigma"]).max())


def test_regression_end_to_end_config1():
    """BASELINE config 1: examples/regression_model, N=1000, fixed schedule (defaults smc_main.jl:123-140).
    Exact log-MDD -99.88901084799365 and posterior moments."""
    spec = models.regression_spec()
    m = models.oracle_model(spec)
    lm, means = [], []
    for seed in (1793, 1794, 1795):
        P0 = orc.initial_draw(m, 1000, seed=seed)
        r = orc.smc_run(m, P0, seed=seed, n_threads=4)
        assert r["n_stages"] == 300 and len(r["ess"]) == 300
        assert r["W"].shape == (1000, 300)
        lm.append(r["logmdd"])
        means.append(orc.weighted_mean(r["particles"]))
        # log-MDD from the stored histories equals the running sum (formula a-9)
        h = np.sum(np.log(np.sum(r["w"][:, 1:] * r["W"][:, :-1], axis=0) / 1000))
        assert h == pytest.approx(r["logmdd"], abs=1e-9)
    assert np.mean(lm) == pytest.approx(-99.88901084799365, abs=0.15)
    np.testing.assert_allclose(np.mean(means, 0), [1.00018685, 0.99936133], atol=0.06)


def test_gauss10_adaptive_end_to_end():
    spec = models.gauss_spec()
    m = models.oracle_model(spec)
    P0 = orc.initial_draw(m, 4000, seed=11)
    r = orc.smc_run(m, P0, seed=11, use_fixed_schedule=False, tempering_target=0.97, n_threads=4, history=False)
    assert r["schedule"][-1] == 1.0 and np.all(np.diff(r["schedule"]) > 0)
    assert r["logmdd"] == pytest.approx(models.gauss_logmdd(), abs=0.25)
    assert models.gauss_logmdd() == pytest.approx(-25.377527143147727, abs=1e-12)
    mj = -1.0 + 2.0 * np.arange(10) / 9
    np.testing.assert_allclose(orc.weighted_mean(r["particles"]), mj * 25 / 25.0625, atol=0.03)
    # ESS drops by the tempering target each non-resampled stage (helpers.jl:14-20)
    ess, res = r["ess"], r["resampled"]
    for k in range(2, len(ess) - 1):
        base = 4000.0 if res[k - 1] else ess[k - 1]
        assert ess[k] == pytest.approx(0.97 * base, rel=1e-6)


def test_oracle_tempered_update_cloud_branches():
    """Oracle restatement of src/smc_main.jl:244-333 (no golden vectors exist for it: structural checks)."""
    data = models.regression_spec()["lik"][2]
    old = np.ascontiguousarray(data[:60])
    sp = dict(priors=[("normal", 0.0, 10.0)] * 2, bounds=[(-1e5, 1e5)] * 2, fixed=[0, 0],
              lik=("linreg", [1.0], data, None), old_lik=("linreg", [1.0], old, None))
    m = models.oracle_model(sp)
    m_old = models.oracle_model(dict(sp, lik=sp["old_lik"], old_lik=None))
    r = orc.smc_run(m_old, orc.initial_draw(m_old, 500, seed=2), n_phi=30, seed=2)
    P_old, ess_old = r["particles"], r["ess"][-1]
    # same size, no prior weight: likelihood columns re-evaluated, θ / weights untouched, ESS carried over
    P, e0 = orc.tempered_update_cloud(m, P_old, ess_old, 500)
    assert e0 == ess_old
    np.testing.assert_array_equal(P[:, [0, 1, 5, 6]], P_old[:, [0, 1, 5, 6]])
    np.testing.assert_array_equal(P[:, 4], P_old[:, 2])
    assert np.all(P[:, 2] < P[:, 4])                       # 100 observations fit worse than 60
    # bridge: 70 % resampled old particles, 30 % prior draws, everything resampled once more -> weights 1, ESS = N
    P, e0 = orc.tempered_update_cloud(m, P_old, ess_old, 400, prior_weight=0.3)
    assert e0 == 400.0 and P.shape == (400, 7)
    np.testing.assert_array_equal(P[:, 6], 1.0)
    assert np.all(np.isfinite(P[:, 2])) and np.all(np.isfinite(P[:, 4]))
    # different size only: every new particle is one of the old ones
    P, _ = orc.tempered_update_cloud(m, P_old, ess_old, 800)
    assert set(map(tuple, P[:, :2])) <= set(map(tuple, P_old[:, :2]))
    # run continues from it
    r2 = orc.smc_run(m, P, n_phi=30, seed=2, initial_ess=800.0)
    assert r2["ess"][0] == 800.0 and np.isfinite(r2["logmdd"])


def test_oracle_kalman_loglik_vs_numpy():
    """Config 5's likelihood has no reference source (parity unpinned): the oracle's filter is checked against an
    independent numpy.linalg statement of the Kalman recursions at random parameter values."""
    sp = models.kalman_spec()
    m = models.oracle_model(sp)
    y = sp["lik"][2]
    rs = np.random.RandomState(0)
    for _ in range(20):
        th = np.concatenate([rs.uniform(-0.9, 0.9, 8), rs.uniform(0.05, 1.5, 4), rs.normal(0, 2, 1)])
        ll = orc.loglik(m.lik, th)
        assert ll == pytest.approx(models.kalman_loglik_numpy(th, y), rel=1e-9, abs=1e-8)
    # the truth explains the data better than a perturbed parameter vector
    assert orc.loglik(m.lik, models.KALMAN_TRUTH) > orc.loglik(m.lik, models.KALMAN_TRUTH * 0.5)
