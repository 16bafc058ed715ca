"""Cloud files and cloud utilities of the host mirror (src/particle.jl:542-648, 705-760; src/smc_main.jl:521-525): no GPU needed."""
import os
import numpy as np
import pytest
from scipy import stats

from tests import models


def _fake_cloud(S, n, d, seed=0):
    rs = np.random.RandomState(seed)
    c = S.Cloud(d, n)
    c.particles = np.asfortranarray(rs.randn(n, d + 5))
    c.particles[:, d + 4] = rs.rand(n)
    c.tempering_schedule = np.linspace(0, 1, 7) ** 2
    c.ESS = rs.rand(7) * n
    c.stage_index, c.n_Phi, c.resamples, c.c, c.accept, c.total_sampling_time = 7, 7, 2, 0.37, 0.21, 1.5
    return c, rs.rand(n, 7), rs.rand(n, 7)


def test_split_and_join_cloud_round_trip(tmp_path):
    import smc_jl_amd as S

    c, w, W = _fake_cloud(S, 120, 3)
    path = str(tmp_path / "smc_cloud.npz")
    S.save_cloud(path, c, w, W)
    S.split_cloud(path, 4)
    for i in range(1, 5):
        part, wp, Wp = S.load_cloud(str(tmp_path / ("smc_cloud_part%d.npz" % i)))
        rows = slice((i - 1) * 30, i * 30)                       # ((i-1)*npart_small+1):(i*npart_small), particle.jl:558
        np.testing.assert_array_equal(part.particles, c.particles[rows])
        np.testing.assert_array_equal(wp, w[rows])
        np.testing.assert_array_equal(Wp, W[rows])
        np.testing.assert_array_equal(part.ESS, c.ESS)            # whole-cloud paths and scalars ride along, :566-573
        assert (part.stage_index, part.resamples, part.c, part.accept) == (7, 2, 0.37, 0.21)
    (tmp_path / "smc_cloud.npz").unlink()
    j, wj, Wj = S.join_cloud(path, 4)
    np.testing.assert_array_equal(j.particles, c.particles)
    np.testing.assert_array_equal(wj, w)
    np.testing.assert_array_equal(Wj, W)
    np.testing.assert_array_equal(j.tempering_schedule, c.tempering_schedule)
    assert (j.stage_index, j.n_Phi, j.total_sampling_time) == (7, 7, 1.5)
    back, wb, Wb = S.load_cloud(path)                             # save_cloud = true wrote the joined file
    np.testing.assert_array_equal(back.particles, c.particles)
    np.testing.assert_array_equal(S.get_cloud(path).ESS, c.ESS)   # src/util.jl:113-115
    with pytest.raises(AssertionError):
        S.split_cloud(path, 7)                                    # @assert mod(n_part, n_pieces) == 0


def test_prior_densities_match_scipy_and_oracle():
    import smc_jl_amd as S
    from smc_jl_amd.host import cloudio
    from oracle import oracle as orc

    cases = [(S.Normal(0.3, 2.0), stats.norm(0.3, 2.0), [-1.0, 0.3, 4.0]),
             (S.Uniform(-1.0, 3.0), stats.uniform(-1.0, 4.0), [-0.5, 2.9]),
             (S.Gamma(2.5, 0.7), stats.gamma(2.5, scale=0.7), [0.1, 1.0, 5.0]),
             (S.Beta(2.0, 3.5), stats.beta(2.0, 3.5), [0.05, 0.5, 0.95]),
             (S.InverseGamma(3.0, 1.2), stats.invgamma(3.0, scale=1.2), [0.2, 1.0, 3.0])]
    for prior, ref, xs in cases:
        for x in xs:
            assert cloudio.prior_logpdf(prior, x) == pytest.approx(ref.logpdf(x), abs=1e-12)
    assert cloudio.prior_logpdf(S.Uniform(0.0, 1.0), 1.5) == -np.inf
    # RootInverseGamma has no scipy twin: the oracle's restatement of ModelConstructors' density
    pri = [("rootinvgamma", 4.0, 0.5), ("normal", 0.0, 1.0)]
    m = orc.Model(pri, [(1e-8, 1e5), (-1e5, 1e5)], orc.Lik("none"), orc.Lik("none"), [0, 0])
    pars = [S.parameter("s", 0.5, (1e-8, 1e5), prior=S.RootInverseGamma(4.0, 0.5)), S.parameter("m", 0.0, prior=S.Normal(0.0, 1.0))]
    for th in ([0.3, 0.1], [1.7, -2.0]):
        assert cloudio.logprior(pars, th) == pytest.approx(orc.logprior(m, np.array(th)), abs=1e-12)


def test_add_parameters_to_cloud_layout():
    """src/particle.jl:705-760: old posterior draws in their places, prior draws for the new parameters, loglh / accept / weight
    columns copied, old_loglh = 0, logprior of the full vector, ESS path of the old cloud, fresh scalars."""
    import smc_jl_amd as S
    from smc_jl_amd.host import cloudio

    old, _, _ = _fake_cloud(S, 200, 2, seed=3)
    pars = [S.parameter("a", 0.0, prior=S.Normal(0.0, 10.0)),
            S.parameter("new1", 0.5, (0.0, 1.0), prior=S.Beta(2.0, 2.0)),
            S.parameter("b", 0.0, prior=S.Normal(0.0, 10.0)),
            S.parameter("new2", 1.0, (1e-6, 50.0), prior=S.Gamma(2.0, 1.5)),
            S.parameter("fixed", 0.25, fixed=True)]
    mask = np.array([True, False, True, False, False])
    c = S.add_parameters_to_cloud(old, pars, mask, seed=5)
    assert c.particles.shape == (200, 5 + 5)
    np.testing.assert_array_equal(c.particles[:, [0, 2]], old.particles[:, :2])
    assert np.all((c.particles[:, 1] > 0.0) & (c.particles[:, 1] < 1.0))
    assert np.all((c.particles[:, 3] > 1e-6) & (c.particles[:, 3] < 50.0))
    np.testing.assert_array_equal(c.particles[:, 4], 0.25)
    np.testing.assert_array_equal(c.particles[:, 5], old.particles[:, 2])          # loglh of the old model
    np.testing.assert_array_equal(c.particles[:, 7], 0.0)                          # old_loglh
    np.testing.assert_array_equal(c.particles[:, 8:], old.particles[:, 5:])        # accept, weight
    for i in (0, 17, 199):
        assert c.particles[i, 6] == pytest.approx(cloudio.logprior(pars, c.particles[i, :5]), abs=0)
    np.testing.assert_array_equal(c.ESS, old.ESS)
    assert (c.stage_index, c.c, c.accept, c.resamples) == (1, 0.0, 0.25, 0)
    # the draws of the new parameters follow their priors (moments, 200 draws)
    assert abs(c.particles[:, 1].mean() - 0.5) < 0.08 and abs(c.particles[:, 3].mean() - 3.0) < 0.6
    # deterministic in the seed
    c2 = S.add_parameters_to_cloud(old, pars, mask, seed=5)
    np.testing.assert_array_equal(c2.particles, c.particles)
    with pytest.raises(ValueError):
        S.add_parameters_to_cloud(old, pars, np.array([True, False, False, False, False]))


def test_hdf5_outputs_round_trip(tmp_path):
    """`particle_store_path` = HDF5 dataset "smcparams" and `savepath` with the reference's .jld2 / .h5 names (src/smc_main.jl:513-526)
    through the built-in minimal HDF5 writer: exact paths (no appended extension), contents back through load_cloud, and - where
    an HDF5 library is importable - through h5py with the Julia array convention (dims reversed, column-major bytes)."""
    from smc_jl_amd.host import cloudio, h5min
    from smc_jl_amd.host.api import Cloud

    rng = np.random.default_rng(3)
    n, d, ns = 50, 3, 7
    c = Cloud(d, n)
    c.particles = np.asfortranarray(rng.normal(size=(n, d + 5)))
    c.tempering_schedule, c.ESS = np.linspace(0, 1, ns), rng.uniform(10, n, ns)
    c.stage_index, c.n_Phi, c.resamples, c.c, c.accept, c.total_sampling_time = ns, 300, 2, 0.31, 0.24, 1.5
    w, W = rng.uniform(size=(n, ns)), rng.uniform(size=(n, ns))
    for name in ("run.jld2", "run.h5", "run.npz", "run.weird"):
        path = str(tmp_path / name)
        cloudio.save_cloud(path, c, w, W)
        assert os.path.exists(path) and not os.path.exists(path + ".npz")           # the exact path is honoured
        c2, w2, W2 = cloudio.load_cloud(path)
        np.testing.assert_array_equal(c2.particles, c.particles)
        np.testing.assert_array_equal(w2, w)
        np.testing.assert_array_equal(W2, W)
        assert (c2.stage_index, c2.n_Phi, c2.resamples, c2.c, c2.accept) == (ns, 300, 2, 0.31, 0.24)
    store = str(tmp_path / "smcsave.h5")
    cloudio.save_smcparams(store, c.particles, d)
    back = h5min.read(store, julia=True)["smcparams"]
    np.testing.assert_array_equal(back, c.particles[:, :d])
    with open(store, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(store, "r") as f:
            np.testing.assert_array_equal(np.asarray(f["smcparams"]).T, c.particles[:, :d])      # HDF5.jl reads it as N x d
        with h5py.File(str(tmp_path / "run.jld2"), "r") as f:
            np.testing.assert_array_equal(np.asarray(f["cloud/particles"]).T, c.particles)
            assert int(np.asarray(f["cloud/stage_index"])) == ns
