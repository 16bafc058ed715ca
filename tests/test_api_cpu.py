"""Host-side mirror of the reference interface (Cloud accessors, argument validation) - CPU only."""
import numpy as np
import pytest


def test_cloud_accessors_match_oracle_moments():
    import smc_jl_amd as S
    from oracle import oracle as orc

    rng = np.random.default_rng(0)
    c = S.Cloud(4, 300)
    c.particles[:] = rng.normal(size=(300, 9))
    c.particles[:, 8] = rng.random(300) + 0.1
    assert len(c) == 300 and not S.cloud_isempty(c) and S.cloud_isempty(S.Cloud(0, 0))
    assert S.get_vals(c).shape == (4, 300) and S.get_vals(c, transpose=False).shape == (300, 4)
    np.testing.assert_array_equal(S.get_loglh(c), c.particles[:, 4])
    np.testing.assert_array_equal(S.get_logprior(c), c.particles[:, 5])
    np.testing.assert_array_equal(S.get_old_loglh(c), c.particles[:, 6])
    np.testing.assert_array_equal(S.get_accept(c), c.particles[:, 7])
    np.testing.assert_array_equal(S.get_weights(c), c.particles[:, 8])
    np.testing.assert_array_equal(S.get_logpost(c), c.particles[:, 4] + c.particles[:, 5])
    np.testing.assert_allclose(S.weighted_mean(c), orc.weighted_mean(c.particles), rtol=1e-12)
    np.testing.assert_allclose(S.weighted_cov(c), orc.weighted_cov(c.particles), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(S.weighted_std(c), np.sqrt(np.diag(orc.weighted_cov(c.particles))), rtol=1e-10)
    # Cloud(n_params, n_parts) defaults, src/particle.jl:50-53
    assert c.stage_index == 1 and c.resamples == 0 and c.accept == 0.25 and c.c == 0.0


def test_smc_argument_errors():
    import smc_jl_amd as S

    pars = [S.parameter("a", 0.0, (-1e5, 1e5), prior=S.Normal(0, 10)), S.parameter("b", 0.0, (-1e5, 1e5), prior=S.Normal(0, 10))]
    data = np.zeros((10, 2))
    with pytest.raises(ValueError, match="Invalid resampler"):
        S.smc(S.LinReg(1.0), pars, data, resampling_method="stratified", verbose="none")
    with pytest.raises(ValueError, match="tempered_update_prior_weight"):
        S.smc(S.LinReg(1.0), pars, data, tempered_update_prior_weight=1.5, verbose="none")
    fixed = [S.parameter("a", 0.0, fixed=True), S.parameter("b", 1.0, fixed=True)]
    with pytest.raises(AssertionError, match="All model parameters are fixed"):
        S.smc(S.LinReg(1.0), fixed, data, verbose="none")
    with pytest.raises(ValueError):
        S.parameter("c", 0.0)                       # free parameter without a prior


def test_host_schedule_and_c_update_match_oracle():
    from oracle import oracle as orc
    from smc_jl_amd.host import hostmath as hm

    s = hm.schedule(100, 2.0)
    np.testing.assert_allclose(s, (np.arange(100) / 99.0) ** 2.0, rtol=1e-15)
    for a in (0.1, 0.25, 0.6):
        assert hm.update_c(0.4, a, 0.25) == pytest.approx(orc.update_c(0.4, a, 0.25), rel=1e-15)
    assert hm.uniform_pair(1234, 77, 5, hm.rng_tag(hm.P_MUT, 3, 1))[0] == pytest.approx(
        __import__("ctypes").c_double(0).value + _orc_uniform(1234, 77, 5, hm.rng_tag(hm.P_MUT, 3, 1)), rel=0, abs=0)


def _orc_uniform(seed, pid, stage, tag):
    import ctypes as C

    from oracle import oracle as orc

    ua, ub = C.c_double(), C.c_double()
    orc.lib().orc_uniform_pair(C.c_uint64(seed), C.c_uint64(pid), C.c_uint32(stage), C.c_uint32(tag), C.byref(ua), C.byref(ub))
    return ua.value
