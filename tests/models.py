"""Model specifications shared by the oracle tests and the HIP parity tests.

A spec is a plain dict: priors [(family, a, b)], bounds [(lo, hi)], fixed [0/1],
lik / old_lik = (family, par, data, aux).  Sources: the reference's example scripts and test
model (cited per function); the 10-dim Gaussian is SURVEY.md §8(d) config 2.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


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


def oracle_model(spec):
    from oracle import oracle as orc

    def mk(l):
        return orc.Lik("none") if l is None else orc.Lik(l[0], l[1], l[2], l[3])

    return orc.Model(spec["priors"], spec["bounds"], mk(spec["lik"]), mk(spec["old_lik"]), spec["fixed"])
