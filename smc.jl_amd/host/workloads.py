"""Workload specifications of the BASELINE configs: what bench.py, the examples and the tests hand to Engine.set_model.

A spec is a plain dict: priors [(family, a, b)], bounds [(lo, hi)], fixed [0/1], lik / old_lik = (family, par, data, aux).
Sources: the reference's example scripts and test model (cited per function); the 10-dim Gaussian is config 2; the state-space model of
config 5 is build-defined (no reference source).  The data files under data/ are the reference's committed example inputs
(examples/regression_model/save/input_data/reg_data.jld2, examples/capm_model/save/input_data/capm.jld2, test/reference/test_data.h5),
extracted by tests/golden/make_fixtures.py.
"""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _g(name):
    return np.load(os.path.join(DATA, name + ".npz"))


def regression_spec():
    """examples/regression_model/estimate_regression.jl:9-10,46-53 (config 1)."""
    data = _g("reg_data")["data"]  # 100 x 2 = [y X]
    return dict(priors=[("normal", 0.0, 10.0)] * 2, bounds=[(-1e5, 1e5)] * 2, fixed=[0, 0],
                lik=("linreg", [1.0], data, None), old_lik=None)


def gauss_spec(d=10, sigma=0.25, prior_sd=5.0):
    """isotropic Gaussian log-likelihood, m_j = -1 + 2 j/(d-1)."""
    m = (-1.0 + 2.0 * np.arange(d) / max(d - 1, 1)).reshape(d, 1)
    return dict(priors=[("normal", 0.0, prior_sd)] * d, bounds=[(-1e5, 1e5)] * d, fixed=[0] * d,
                lik=("gauss_iso", [sigma], m, None), old_lik=None)


def gauss_logmdd(d=10, sigma=0.25, prior_sd=5.0):
    m = -1.0 + 2.0 * np.arange(d) / max(d - 1, 1)
    v = sigma ** 2 + prior_sd ** 2
    return float(np.sum(-0.5 * np.log(2 * np.pi * v) - m ** 2 / (2 * v)))


def linmodel_spec(T=100, old_T=None, prior_para=1e3):
    """test/modelsetup.jl:9-67 (9 parameters) + loglik_fn :119-138; data/X from test_data.h5."""
    z = _g("linmodel")
    data, X = z["data"], z["X"]
    pri, bnd = [], []
    for _ in range(3):
        pri += [("normal", 0.0, prior_para), ("normal", 0.0, prior_para), ("uniform", 0.0, prior_para)]
        bnd += [(-1e5, 1e5), (-1e5, 1e5), (1e-5, 1e5)]
    old = None if old_T is None else ("linmodel3", [], data[:, :old_T], X)
    return dict(priors=pri, bounds=bnd, fixed=[0] * 9, lik=("linmodel3", [], data[:, :T], X), old_lik=old)


def capm_spec():
    """examples/capm_model/estimate_capm.jl:16-33,52-70 (config 4, literal likelihood)."""
    z = _g("capm_data")
    pri, bnd = [], []
    for _ in range(3):
        pri += [("normal", 0.0, 1e3), ("normal", 0.0, 1e3), ("uniform", 0.0, 1e3)]
        bnd += [(-1e5, 1e5), (-1e5, 1e5), (1e-5, 1e5)]
    return dict(priors=pri, bounds=bnd, fixed=[0] * 9, lik=("capm_literal", [], z["lik_data"], z["market_data"]),
                old_lik=None)


def kalman_structure():
    """Fixed structure of the config-5 state-space model (build-defined): coupling pattern C (8x8),
    shock loadings R (8x3), measurement Z (3x8), packed row-major as the `aux` block of the lgss_kalman family."""
    i, j = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    C = 0.5 * np.cos(1.0 + i + 2.0 * j) * (i != j)
    R = np.zeros((8, 3))
    for k in range(8):
        R[k, k % 3] = 1.0 / (1.0 + k // 3)
    a, k = np.meshgrid(np.arange(3), np.arange(8), indexing="ij")
    Z = 1.0 / (1.0 + np.abs(k - 3 * a))
    return C, R, Z


KALMAN_TRUTH = np.array([0.9, 0.7, 0.5, 0.3, -0.2, 0.6, 0.8, 0.4, 0.5, 0.3, 0.2, 0.25, 1.0])
KALMAN_KAPPA = 0.2


def kalman_data(T=80, seed=123):
    """Synthetic observations from the model at KALMAN_TRUTH (numpy legacy RandomState: stable across versions)."""
    C, R, Z = kalman_structure()
    th = KALMAN_TRUTH
    Tm = np.diag(th[:8]) + KALMAN_KAPPA * C
    rs = np.random.RandomState(seed)
    x = np.zeros(8)
    y = np.zeros((3, T))
    for t in range(T):
        x = Tm @ x + R @ (th[8:11] * rs.standard_normal(3))
        y[:, t] = th[12] + Z @ x + th[11] * rs.standard_normal(3)
    return y


def kalman_loglik_numpy(th, y, kappa=KALMAN_KAPPA):
    """Textbook Kalman-filter log-likelihood with numpy.linalg (independent of the oracle's statement order)."""
    C, R, Z = kalman_structure()
    Tm = np.diag(th[:8]) + kappa * C
    Q = R @ np.diag(th[8:11] ** 2) @ R.T
    E = th[11] ** 2 * np.eye(3)
    x, P, ll = np.zeros(8), np.eye(8), 0.0
    for t in range(y.shape[1]):
        x = Tm @ x
        P = Tm @ P @ Tm.T + Q
        v = y[:, t] - th[12] - Z @ x
        F = Z @ P @ Z.T + E
        Fi = np.linalg.inv(F)
        ll += -1.5 * np.log(2 * np.pi) - 0.5 * np.log(np.linalg.det(F)) - 0.5 * v @ Fi @ v
        K = P @ Z.T @ Fi
        x = x + K @ v
        P = P - K @ Z @ P
    return ll


def kalman_spec(T=80, old_T=None):
    """Config 5: 13 parameters = 8 AR coefficients, 3 shock std, measurement std, measurement mean."""
    y = kalman_data(80)
    C, R, Z = kalman_structure()
    aux = np.concatenate([C.ravel(), R.ravel(), Z.ravel()]).reshape(1, -1)
    pri = [("uniform", -0.95, 0.95)] * 8 + [("uniform", 0.0, 2.0)] * 4 + [("normal", 0.0, 5.0)]
    bnd = [(-0.95, 0.95)] * 8 + [(1e-3, 2.0)] * 4 + [(-1e5, 1e5)]
    old = None if old_T is None else ("lgss_kalman", [KALMAN_KAPPA], np.ascontiguousarray(y[:, :old_T]), aux)
    return dict(priors=pri, bounds=bnd, fixed=[0] * 13, lik=("lgss_kalman", [KALMAN_KAPPA], np.ascontiguousarray(y[:, :T]), aux),
                old_lik=old)


def gauss_closures(spec, tempered=False):
    """The gauss_iso likelihood of `spec` as HOST closures theta (m, d) -> (m,) - what a user hands to smc(loglikelihood, ...)
    (src/smc_main.jl:118); tempered: a second, wider one as `old_loglikelihood` (generalized tempering, src/mutation.jl:96-106)."""
    m, sigma = np.asarray(spec["lik"][2]).ravel(), float(spec["lik"][1][0])
    d = m.size

    def lik(theta, s=sigma):
        return -0.5 * d * np.log(2.0 * np.pi * s * s) - 0.5 * (((theta - m) / s) ** 2).sum(axis=1)

    fns = [lik]
    if tempered:
        fns.append(lambda theta: lik(theta, 3.0 * sigma))
    return dict(fns=fns)
