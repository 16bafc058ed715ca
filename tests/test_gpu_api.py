"""smc(loglikelihood, parameters, data; ...) mirror on the GPU: device likelihood families and arbitrary host callables."""
import numpy as np
import pytest

from tests import models

pytestmark = pytest.mark.gpu


def _reg_parameters(S):
    return [S.parameter("α1", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False),
            S.parameter("β1", 0.0, (-1e5, 1e5), (-1e5, 1e5), None, S.Normal(0, 10), fixed=False)]


def test_smc_regression_device_likelihood(tmp_path):
    """examples/regression_model/estimate_regression.jl through the mirror API (config 1 defaults, N = 1000)."""
    import smc_jl_amd as S
    from oracle import oracle as orc

    data = models.regression_spec()["lik"][2]
    sp = str(tmp_path / "smc_cloud.npz")
    cloud, w, W = S.smc(S.LinReg(1.0), _reg_parameters(S), data, n_parts=1000, use_fixed_schedule=True, seed=1793,
                        verbose="none", savepath=sp)
    assert cloud.stage_index == 300 and len(cloud.ESS) == 300 and len(cloud.tempering_schedule) == 300
    assert w.shape == W.shape == (1000, 300)
    np.testing.assert_allclose(cloud.tempering_schedule, (np.arange(300) / 299.0) ** 2.1, rtol=1e-14)
    assert cloud.accept == pytest.approx(S.get_accept(cloud).mean(), rel=1e-12)       # update_acceptance_rate!
    # identical to the oracle loop on the same seed
    m = models.oracle_model(models.regression_spec())
    r = orc.smc_run(m, orc.initial_draw(m, 1000, seed=1793), seed=1793, n_threads=2)
    assert cloud.logmdd == pytest.approx(r["logmdd"], abs=1e-6)
    np.testing.assert_allclose(cloud.ESS, r["ess"], rtol=1e-6)
    z = np.load(sp)
    np.testing.assert_array_equal(z["particles"], cloud.particles)
    np.testing.assert_allclose(S.weighted_mean(cloud), [1.0, 1.0], atol=0.2)


def test_smc_host_callable_equals_device_family():
    """An arbitrary Python log-likelihood (the Julia-closure case) through propose/accept reproduces the fused device path."""
    import smc_jl_amd as S

    data = models.regression_spec()["lik"][2]
    y, X = data[:, 0], data[:, 1]

    def loglik(p, d):
        e = d[:, 0] - p[0] - p[1] * d[:, 1]
        return -(len(y) / 2) * np.log(2 * np.pi) - (len(y) / 2) * np.log(1.0) - 0.5 * np.dot(e, e)

    kw = dict(n_parts=400, n_phi=25, use_fixed_schedule=True, seed=7, verbose="none", n_blocks=2, alpha=0.9)
    c1, w1, W1 = S.smc(S.LinReg(1.0), _reg_parameters(S), data, **kw)
    c2, w2, W2 = S.smc(loglik, _reg_parameters(S), data, **kw)
    assert c1.stage_index == c2.stage_index == 25 and c1.resamples == c2.resamples
    np.testing.assert_allclose(c2.ESS, c1.ESS, rtol=1e-6)
    assert c2.logmdd == pytest.approx(c1.logmdd, abs=1e-6)
    same = c1.particles[:, 5] == c2.particles[:, 5]
    assert same.mean() > 0.98
    np.testing.assert_allclose(c2.particles[same][:, :4], c1.particles[same][:, :4], rtol=1e-7, atol=1e-9)


def test_smc_adaptive_gauss_returns_consistent_cloud():
    import smc_jl_amd as S

    d = 5
    m = (-1.0 + 2.0 * np.arange(d) / (d - 1)).reshape(d, 1)
    pars = [S.parameter("t%d" % k, 0.0, (-1e5, 1e5), prior=S.Normal(0, 5)) for k in range(d)]
    cloud, w, W = S.smc(S.GaussIso(0.25), pars, m, n_parts=20000, use_fixed_schedule=False, tempering_target=0.95, seed=3,
                        verbose="none")
    assert cloud.tempering_schedule[-1] == 1.0 and np.all(np.diff(cloud.tempering_schedule) > 0)
    assert cloud.logmdd == pytest.approx(models.gauss_logmdd(d), abs=0.1)
    h = np.sum(np.log(np.sum(w[:, 1:] * W[:, :-1], axis=0) / 20000))
    assert h == pytest.approx(cloud.logmdd, abs=1e-8)
    np.testing.assert_allclose(S.weighted_mean(cloud), m.ravel() * 25 / 25.0625, atol=0.02)
