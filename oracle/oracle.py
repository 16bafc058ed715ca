"""ctypes loader for the CPU ORACLE (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (smc.jl_amd -> libsmcmi.so) never imports this module.

Clouds are numpy float64 arrays of shape (N, R) in FORTRAN order (= the reference's
`cloud.particles`, Julia column-major; src/particle.jl:31-63).
"""
import copy
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PRIOR = {"normal": 0, "uniform": 1, "gamma": 2, "beta": 3, "invgamma": 4, "rootinvgamma": 5}
LIK = {"gauss_iso": 0, "linreg": 1, "linmodel3": 2, "capm_literal": 3, "lgss_kalman": 4, "none": -1}
RESAMPLE = {"systematic": 0, "multinomial": 1, "polyalgo": 1}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class _Lik(C.Structure):
    _fields_ = [("family", C.c_int32), ("par", _dp), ("n_par", C.c_int64), ("data", _dp), ("rows", C.c_int64),
                ("cols", C.c_int64), ("aux", _dp), ("aux_rows", C.c_int64), ("aux_cols", C.c_int64)]


class _Model(C.Structure):
    _fields_ = [("n_para", C.c_int32), ("fixed", _ip), ("lo", _dp), ("hi", _dp), ("prior_family", _ip),
                ("prior_a", _dp), ("prior_b", _dp), ("lik", _Lik), ("old_lik", _Lik)]


class _RunConfig(C.Structure):
    _fields_ = [("n_parts", C.c_int64), ("n_blocks", C.c_int32), ("n_mh_steps", C.c_int32), ("lam", C.c_double),
                ("n_phi", C.c_int32), ("resampling_method", C.c_int32), ("threshold_ratio", C.c_double),
                ("c", C.c_double), ("alpha", C.c_double), ("target", C.c_double), ("use_fixed_schedule", C.c_int32),
                ("tempering_target", C.c_double), ("prior_weight", C.c_double), ("log_prob_old_data", C.c_double),
                ("seed", C.c_uint64), ("max_stages", C.c_int32), ("n_threads", C.c_int32), ("initial_ess", C.c_double),
                ("variant", C.c_int32), ("pad_", C.c_int32)]


class _RunResult(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("resamples", C.c_int32), ("logmdd", C.c_double), ("c", C.c_double),
                ("accept", C.c_double), ("seconds", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = [os.path.join(_HERE, f) for f in ("smc_oracle.c", "smc_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_compute_ess.restype = C.c_double
        _LIB.orc_update_c.restype = C.c_double
        _LIB.orc_logprior.restype = C.c_double
        _LIB.orc_loglik.restype = C.c_double
        _LIB.orc_last_error.restype = C.c_char_p
        _LIB.orc_compute_ess.argtypes = [_dp, _dp, _dp, C.c_int64, C.c_double, C.c_double]
        _LIB.orc_update_c.argtypes = [C.c_double] * 3
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def fcloud(a):
    """(N, R) float64 in Fortran order (copy)."""
    return np.array(a, dtype=np.float64, order="F", copy=True)


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(lib().orc_last_error().decode())


class Lik:
    """A built-in likelihood family + its data (column-major, Julia layout)."""

    def __init__(self, family, par=(), data=None, aux=None):
        self.family = LIK[family] if isinstance(family, str) else int(family)
        self.par = _f64(par).ravel()
        self.data = None if data is None else np.asfortranarray(np.atleast_2d(np.asarray(data, dtype=np.float64)))
        self.aux = None if aux is None else np.asfortranarray(np.atleast_2d(np.asarray(aux, dtype=np.float64)))

    def struct(self):
        s = _Lik()
        s.family = self.family
        s.par = _d(self.par) if self.par.size else None
        s.n_par = self.par.size
        if self.data is not None:
            s.data, s.rows, s.cols = _d(self.data), self.data.shape[0], self.data.shape[1]
        if self.aux is not None:
            s.aux, s.aux_rows, s.aux_cols = _d(self.aux), self.aux.shape[0], self.aux.shape[1]
        return s


class Model:
    """Parameter vector description (ModelConstructors ParameterVector restated as arrays) + likelihoods."""

    def __init__(self, priors, bounds, lik, old_lik=None, fixed=None):
        d = len(priors)
        self.d = d
        self.prior_family = np.array([PRIOR[p[0]] for p in priors], dtype=np.int32)
        self.prior_a = _f64([p[1] for p in priors])
        self.prior_b = _f64([p[2] for p in priors])
        self.lo = _f64([b[0] for b in bounds])
        self.hi = _f64([b[1] for b in bounds])
        self.fixed = np.zeros(d, dtype=np.int32) if fixed is None else np.asarray(fixed, dtype=np.int32)
        self.lik = lik
        self.old_lik = old_lik if old_lik is not None else Lik("none")
        self.free_inds = np.flatnonzero(self.fixed == 0).astype(np.int32)

    def struct(self):
        m = _Model()
        m.n_para = self.d
        m.fixed, m.lo, m.hi = _i(self.fixed), _d(self.lo), _d(self.hi)
        m.prior_family, m.prior_a, m.prior_b = _i(self.prior_family), _d(self.prior_a), _d(self.prior_b)
        m.lik = self.lik.struct()
        m.old_lik = self.old_lik.struct()
        return m


# ----------------------------------------------------------------------------- function wrappers
def model_from_spec(spec):
    """The oracle's Model for a workload spec (smc.jl_amd/host/workloads.py: priors, bounds, fixed, lik, old_lik)."""
    def mk(l):
        return Lik("none") if l is None else Lik(l[0], l[1], l[2], l[3])

    return Model(spec["priors"], spec["bounds"], mk(spec["lik"]), mk(spec["old_lik"]), spec["fixed"])


def compute_ess(loglh, weights, phi_n, phi_n1, old_loglh=None):
    loglh, weights = _f64(loglh), _f64(weights)
    old = None if old_loglh is None else _f64(old_loglh)
    return lib().orc_compute_ess(_d(loglh), _d(weights), None if old is None else _d(old), loglh.size, phi_n, phi_n1)


def solve_adaptive_phi(particles, ess_prev, sched, j, phi_prop, phi_n1, target, resampled_last):
    p = fcloud(particles)
    sched = _f64(sched)
    jj, pp, rl = C.c_int32(j), C.c_double(phi_prop), C.c_int32(int(resampled_last))
    out, ne = C.c_double(), C.c_int32()
    _check(lib().orc_solve_adaptive_phi(_d(p), C.c_int64(p.shape[0]), C.c_int32(p.shape[1]), C.c_double(ess_prev),
                                        _d(sched), C.c_int32(sched.size), C.byref(jj), C.byref(pp),
                                        C.c_double(phi_n1), C.c_double(target), C.byref(rl), C.byref(out), C.byref(ne)))
    return out.value, bool(rl.value), jj.value, pp.value, ne.value


def proposal_densities(para_draw, para_subset, mu, Sigma, c, alpha):
    pd, ps, mu, S = _f64(para_draw), _f64(para_subset), _f64(mu), _f64(Sigma)
    q0, q1 = C.c_double(), C.c_double()
    lib().orc_proposal_densities(_d(pd), _d(ps), _d(mu), _d(S), C.c_int32(mu.size), C.c_double(c), C.c_double(alpha),
                                 C.byref(q0), C.byref(q1))
    return q0.value, q1.value


def mixture_draw(theta_old, mu, Sigma, c, alpha, seed, pid, stage, t):
    th, mu, S = _f64(theta_old), _f64(mu), _f64(Sigma)
    out = np.empty_like(th)
    _check(lib().orc_mixture_draw(_d(th), _d(mu), _d(S), C.c_int32(mu.size), C.c_double(c), C.c_double(alpha),
                                  C.c_uint64(seed), C.c_uint64(pid), C.c_uint32(stage), C.c_uint32(t), _d(out)))
    return out


def generate_blocks(n_free, n_blocks, free_inds, seed, stage):
    free_inds = np.asarray(free_inds, dtype=np.int32)
    bf, ba = np.empty(n_free, np.int32), np.empty(n_free, np.int32)
    bp = np.empty(n_blocks + 1, np.int32)
    lib().orc_generate_blocks(C.c_int32(n_free), C.c_int32(n_blocks), _i(free_inds), C.c_uint64(seed),
                              C.c_uint32(stage), _i(bf), _i(ba), _i(bp))
    return bf, ba, bp


def correct(particles, phi_n, phi_n1, pw=0.0, logp_old=0.0):
    """Returns (updated cloud, inc_w, norm_w, ess, sum_unnormalised)."""
    p = fcloud(particles)
    n = p.shape[0]
    inc, nw = np.empty(n), np.empty(n)
    ess, su = C.c_double(), C.c_double()
    lib().orc_correct(_d(p), C.c_int64(n), C.c_int32(p.shape[1]), C.c_double(phi_n), C.c_double(phi_n1),
                      C.c_double(pw), C.c_double(logp_old), _d(inc), _d(nw), C.byref(ess), C.byref(su))
    return p, inc, nw, ess.value, su.value


def resample(weights, method="systematic", seed=0, stage=0, n_parts=None, offsets=None):
    w = _f64(weights)
    n_parts = w.size if n_parts is None else n_parts
    idx = np.empty(n_parts, dtype=np.int64)
    m = RESAMPLE[method]
    if offsets is None:
        lib().orc_resample(_d(w), C.c_int64(w.size), C.c_int64(n_parts), C.c_int32(m), C.c_uint64(seed),
                           C.c_uint32(stage), idx.ctypes.data_as(_lp))
    else:
        off = _f64(np.atleast_1d(offsets))
        lib().orc_resample_with_offsets(_d(w), C.c_int64(w.size), C.c_int64(n_parts), C.c_int32(m), _d(off),
                                        idx.ctypes.data_as(_lp))
    return idx


def weighted_mean(particles):
    p = fcloud(particles)
    out = np.empty(p.shape[1] - 5)
    lib().orc_weighted_mean(_d(p), C.c_int64(p.shape[0]), C.c_int32(p.shape[1]), _d(out))
    return out


def weighted_cov(particles):
    p = fcloud(particles)
    d = p.shape[1] - 5
    out = np.empty((d, d))
    lib().orc_weighted_cov(_d(p), C.c_int64(p.shape[0]), C.c_int32(p.shape[1]), _d(out))
    return out


def update_c(c, accept, target):
    return lib().orc_update_c(c, accept, target)


def logprior(model, theta):
    m = model.struct()
    th = _f64(theta)
    return lib().orc_logprior(C.byref(m), _d(th))


def loglik(lik, theta):
    s = lik.struct()
    th = _f64(theta)
    return lib().orc_loglik(C.byref(s), _d(th), C.c_int32(th.size))


def mutate_cloud(model, particles, mu_free, Sigma_free, blocks_free, blocks_all, block_ptr, phi_n, phi_n1, c, alpha,
                 n_mh_steps, seed, stage, pid0=0, n_threads=1):
    p = fcloud(particles)
    m = model.struct()
    mu, S = _f64(mu_free), _f64(Sigma_free)
    bf, ba, bp = (np.asarray(x, dtype=np.int32) for x in (blocks_free, blocks_all, block_ptr))
    _check(lib().orc_mutate_cloud(C.byref(m), _d(p), C.c_int64(p.shape[0]), C.c_int64(pid0), _d(mu), _d(S),
                                  C.c_int32(mu.size), _i(bf), _i(ba), _i(bp), C.c_int32(bp.size - 1),
                                  C.c_double(phi_n), C.c_double(phi_n1), C.c_double(c), C.c_double(alpha),
                                  C.c_int32(n_mh_steps), C.c_uint64(seed), C.c_uint32(stage), C.c_int32(n_threads)))
    return p


def initial_draw(model, n, seed, pid0=0):
    p = np.zeros((n, model.d + 5), order="F")
    m = model.struct()
    _check(lib().orc_initial_draw(C.byref(m), _d(p), C.c_int64(n), C.c_int64(pid0), C.c_uint64(seed)))
    return p


def initialize_likelihoods(model, particles):
    """initialize_likelihoods! (src/initialization.jl:153-186)."""
    p = fcloud(particles)
    m = model.struct()
    lib().orc_initialize_likelihoods(C.byref(m), _d(p), C.c_int64(p.shape[0]))
    return p


def tempered_update_cloud(model, old_particles, old_ess_last, n_parts, prior_weight=0.0, resampling_method="systematic",
                          seed=0):
    """Initial cloud of a tempered update (src/smc_main.jl:244-333).  `model.lik` = new likelihood/data,
    `model.old_lik` = old likelihood/old data.  Returns (particles, ESS[1]).  RNG: bridge resample = stage 0,
    clean-up resample = stage 1, prior draws = initial_draw streams of particle ids 0..n_from_prior-1."""
    old = fcloud(old_particles)
    old_n, R = old.shape
    d = R - 5
    if prior_weight == 0.0 and old_n == n_parts:                      # :249-260
        return initialize_likelihoods(model, old), float(old_ess_last)
    n_to = int(round((1.0 - prior_weight) * n_parts))                  # :262-264
    n_pr = n_parts - n_to
    parts = []
    if n_to > 0:
        idx = resample(old[:, R - 1].copy(), n_parts=n_to, method=resampling_method, seed=seed, stage=0)
        parts.append(old[idx, :])                                      # update_cloud!: whole rows, old weights kept
    if n_pr > 0:
        pm = copy.copy(model)
        pm.lik, pm.old_lik = model.old_lik, Lik("none")
        parts.append(initial_draw(pm, n_pr, seed))                     # :288-291 old_loglikelihood on old_data
    p = np.asfortranarray(np.vstack(parts))
    p = initialize_likelihoods(model, p)                               # :308
    w = p[:, R - 1]
    w[p[:, d] == -np.inf] = 0.0                                        # zero_bad_loglh_weights! :313
    sw = 0.0
    for v in w:                                                        # sequential sum as Julia's sum over a column view
        sw += v
    p[:, R - 1] = (w * n_parts) / sw                                   # normalize_weights! :314
    idx = resample(p[:, R - 1] / n_parts, n_parts=n_parts, method=resampling_method, seed=seed, stage=1)   # :317
    p = np.asfortranarray(p[idx, :])
    p[:, R - 1] = 1.0                                                  # reset_weights! :322
    return p, float(n_parts)                                           # push!(cloud.ESS, n_parts) :325


def smc_run(model, particles, n_blocks=1, n_mh_steps=1, lam=2.1, n_phi=300, resampling_method="systematic",
            threshold_ratio=0.5, c=0.5, alpha=1.0, target=0.25, use_fixed_schedule=True, tempering_target=0.97,
            prior_weight=0.0, log_prob_old_data=0.0, seed=0, max_stages=None, n_threads=1, history=True, initial_ess=0.0,
            variant=0):
    """The reference's while-loop (src/smc_main.jl:377-508) on an initial cloud.  Returns a dict."""
    p = fcloud(particles)
    n = p.shape[0]
    if max_stages is None:
        max_stages = n_phi if use_fixed_schedule else 20 * n_phi
    cfg = _RunConfig(n, n_blocks, n_mh_steps, lam, n_phi, RESAMPLE[resampling_method], threshold_ratio, c, alpha,
                     target, int(use_fixed_schedule), tempering_target, prior_weight, log_prob_old_data, seed,
                     max_stages, n_threads, initial_ess, int(variant), 0)
    sched, ess, cs, acc = (np.zeros(max_stages) for _ in range(4))
    res_flags = np.zeros(max_stages, dtype=np.int32)
    wh = Wh = None
    if history:
        wh, Wh = np.zeros((n, max_stages), order="F"), np.zeros((n, max_stages), order="F")
    res = _RunResult()
    m = model.struct()
    rc = lib().orc_smc_run(C.byref(m), C.byref(cfg), _d(p), _d(sched), _d(ess), _d(cs), _d(acc), _i(res_flags),
                           None if wh is None else _d(wh), None if Wh is None else _d(Wh), C.byref(res))
    _check(rc)
    s = res.n_stages
    out = dict(particles=p, n_stages=s, resamples=res.resamples, logmdd=res.logmdd, c=res.c, accept=res.accept,
               seconds=res.seconds, schedule=sched[:s].copy(), ess=ess[:s].copy(), c_hist=cs[:s].copy(),
               accept_hist=acc[:s].copy(), resampled=res_flags[:s].copy())
    if history:
        out["w"], out["W"] = wh[:, :s], Wh[:, :s]
    return out
