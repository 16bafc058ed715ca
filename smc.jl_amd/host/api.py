"""Python mirror of the reference's user-facing interface for the hot path:

    smc(loglikelihood, parameters, data; kwargs...)          src/smc_main.jl:118-161
    Cloud + get_vals / get_loglh / ... / weighted_mean / weighted_cov / weighted_std   src/particle.jl
    parameter(...) / Normal / Uniform / ...                   ModelConstructors.jl + Distributions.jl (absent deps)

Same names, argument meaning and error behaviour; the work happens in libsmcmi.so (HIP).  A Julia user gets the same
surface through smc.jl_amd/julia/SMCMI.jl (ccall); this module exists because Julia is not installed in the build image,
so the parity tests drive the C ABI from Python.

`loglikelihood` is either a device family (GaussIso, LinReg, LinModel3, CapmLiteral: evaluated inside the mutation
kernel) or any Python callable f(theta: ndarray, data: ndarray) -> float, which runs through the propose / accept split
(device proposal + prior + MH decision, host likelihood) exactly as a Julia closure would.
"""
import math
import os
import time

import numpy as np

from . import hostmath as hm
from ._lib import SMCMIError
from .engine import Engine

VERBOSITY = {"none": 0, "low": 1, "high": 2}


# ----------------------------------------------------------------------------------------------- priors / parameters
class _Prior:
    family = None

    def __init__(self, a, b):
        self.a, self.b = float(a), float(b)

    def triple(self):
        return (self.family, self.a, self.b)


class Normal(_Prior):          # Distributions.Normal(μ, σ)
    family = "normal"


class Uniform(_Prior):         # Distributions.Uniform(a, b)
    family = "uniform"


class Gamma(_Prior):           # Distributions.Gamma(shape, scale)
    family = "gamma"


class Beta(_Prior):            # Distributions.Beta(α, β)
    family = "beta"


class InverseGamma(_Prior):    # Distributions.InverseGamma(shape, scale)
    family = "invgamma"


class RootInverseGamma(_Prior):  # ModelConstructors.RootInverseGamma(ν, τ)
    family = "rootinvgamma"


class Parameter:
    """ModelConstructors.Parameter as far as SMC reads it.  `regimes` (ModelConstructors' regime-switching values, set with
    set_regime_val! / set_regime_prior! / set_regime_fixed! / set_regime_valuebounds!): regime 1 is the parameter itself, regime i >= 2
    is regimes[i - 2] = Parameter-like entry with its own value, prior, bounds and fixedness."""

    def __init__(self, key, value, valuebounds, prior, fixed=False):
        self.key, self.value, self.valuebounds, self.prior, self.fixed = key, float(value), tuple(valuebounds), prior, bool(fixed)
        self.regimes = []

    def add_regime(self, value, prior=None, valuebounds=None, fixed=None):
        """One more regime value of this parameter (defaults: the parameter's own prior / bounds / fixedness)."""
        fx = self.fixed if fixed is None else bool(fixed)
        r = Parameter("%s_reg%d" % (self.key, len(self.regimes) + 2), value, self.valuebounds if valuebounds is None else valuebounds,
                      self.prior if prior is None else prior, fx)
        self.regimes.append(r)
        return self


def flatten_regimes(parameters, regime_switching=True):
    """The parameter vector SMC samples when regime_switching = true (src/smc_main.jl:207-234): the regime-1 entries of all parameters
    in order, then, parameter by parameter, the values of regimes 2, 3, ... - each a column of cloud.particles with its own prior,
    bounds and fixedness (para_symbols: key, ..., key_reg2, ...).  Without regime switching: the parameters themselves."""
    parameters = list(parameters)
    if not regime_switching:
        return parameters
    return parameters + [r for p in parameters for r in getattr(p, "regimes", [])]


def regime_values(parameters, theta):
    """What update!(parameters, theta) leaves in a regime-switching ParameterVector (src/mutation.jl:93): key -> [value in regime 1,
    regime 2, ...] for a flattened draw `theta`.  A likelihood closure uses it to read the regime-dependent parameters."""
    parameters = list(parameters)
    out = {p.key: [float(theta[k])] for k, p in enumerate(parameters)}
    pos = len(parameters)
    for p in parameters:
        for _ in getattr(p, "regimes", []):
            out[p.key].append(float(theta[pos]))
            pos += 1
    return out


def parameter(key, value, valuebounds=(-1e5, 1e5), transform_parameterization=None, transform=None, prior=None, fixed=False):
    """ModelConstructors.parameter(key, value, valuebounds, transform_parameterization, transform, prior; fixed).
    The transform arguments are accepted for call compatibility; SMC itself never uses them."""
    if prior is None and not fixed:
        raise ValueError("a free parameter needs a prior")
    return Parameter(key, value, valuebounds, prior, fixed)


def _spec_from(parameters, lik, old_lik):
    pri, bnd, fx = [], [], []
    for p in parameters:
        if p.fixed:
            pri.append(("normal", p.value, 1.0))        # fixed: the value rides in prior_a (see k_init_prior)
        else:
            pri.append(p.prior.triple())
        bnd.append(p.valuebounds)
        fx.append(1 if p.fixed else 0)
    return dict(priors=pri, bounds=bnd, fixed=fx, lik=lik, old_lik=old_lik)


# ----------------------------------------------------------------------------------------------- device likelihoods
class DeviceLikelihood:
    family = None

    def spec(self, data):
        raise NotImplementedError


class GaussIso(DeviceLikelihood):
    """ℓ(θ) = -(d/2) log(2πσ²) - Σ(θ_j - m_j)²/(2σ²); data = m (d x 1)."""

    def __init__(self, sigma):
        self.sigma = float(sigma)

    def spec(self, data):
        return ("gauss_iso", [self.sigma], np.asarray(data, dtype=np.float64).reshape(-1, 1), None)


class LinReg(DeviceLikelihood):
    """examples/regression_model/estimate_regression.jl:46-53; data = [y X] (n x 2)."""

    def __init__(self, sigma2=1.0):
        self.sigma2 = float(sigma2)

    def spec(self, data):
        return ("linreg", [self.sigma2], np.asarray(data, dtype=np.float64), None)


class LinModel3(DeviceLikelihood):
    """test/modelsetup.jl:119-138 loglik_fn; data 3 x T, regressors X 3 x (>= T)."""

    def __init__(self, X):
        self.X = np.asarray(X, dtype=np.float64)

    def spec(self, data):
        return ("linmodel3", [], np.asarray(data, dtype=np.float64), self.X)


class LGSSKalman(DeviceLikelihood):
    """Linear-Gaussian state-space model, Kalman-filter likelihood per particle (csrc/model.hpp
    kalman_lgss).  8 states, 3 observables, 3 shocks, 13 parameters (ρ[8], σ[3], σ_e, μ); transition diag(ρ) + κ C, shock
    loadings R (8x3), measurement matrix Z (3x8); data 3 x T."""

    def __init__(self, C, R, Z, kappa):
        self.aux = np.concatenate([np.asarray(C, dtype=np.float64).reshape(64), np.asarray(R, dtype=np.float64).reshape(24),
                                   np.asarray(Z, dtype=np.float64).reshape(24)]).reshape(1, -1)
        self.kappa = float(kappa)

    def spec(self, data):
        return ("lgss_kalman", [self.kappa], np.asarray(data, dtype=np.float64), self.aux)


class CapmLiteral(DeviceLikelihood):
    """examples/capm_model/estimate_capm.jl:52-70 as written; data 3 x T, market 1 x T."""

    def __init__(self, market_data):
        self.market = np.asarray(market_data, dtype=np.float64).reshape(1, -1)

    def spec(self, data):
        return ("capm_literal", [], np.asarray(data, dtype=np.float64), self.market)


# ----------------------------------------------------------------------------------------------- Cloud
class Cloud:
    """src/particle.jl:31-53.  particles: (n_parts, n_params + 5) float64, Fortran order."""

    def __init__(self, n_params=0, n_parts=0):
        self.particles = np.empty((n_parts, n_params + 5), order="F")
        self.tempering_schedule = np.zeros(1)
        self.ESS = np.zeros(1)
        self.stage_index = 1
        self.n_Phi = 0
        self.resamples = 0
        self.c = 0.0
        self.accept = 0.25
        self.total_sampling_time = 0.0

    def __len__(self):
        return self.particles.shape[0]


def _P(c):
    return c.particles if isinstance(c, Cloud) else c


def get_vals(c, transpose=True):
    v = _P(c)[:, :-5]
    return np.array(v.T if transpose else v)


def get_loglh(c):
    return np.array(_P(c)[:, -5])


def get_logprior(c):
    return np.array(_P(c)[:, -4])


def get_old_loglh(c):
    return np.array(_P(c)[:, -3])


def get_logpost(c):
    return get_loglh(c) + get_logprior(c)


def get_accept(c):
    return np.array(_P(c)[:, -2])


def get_weights(c):
    return np.array(_P(c)[:, -1])


def weighted_mean(c):
    w = get_weights(c)
    return get_vals(c) @ w / w.sum()


def weighted_cov(c):
    w = get_weights(c)
    w = w / w.sum()
    X = get_vals(c, transpose=False)
    m = (w @ X) / w.sum()
    Xc = X - m
    return (Xc.T * w) @ Xc / w.sum()


def weighted_std(c):
    return np.sqrt(np.diag(weighted_cov(c)))


def cloud_isempty(c):
    return len(c) == 0


# ----------------------------------------------------------------------------------------------- smc()
def smc(loglikelihood, parameters, data, *, verbose="low", n_parts=5000, n_blocks=1, n_mh_steps=1, lam=2.1, n_phi=300,
        resampling_method="systematic", threshold_ratio=0.5, c=0.5, alpha=1.0, target=0.25, use_fixed_schedule=True,
        tempering_target=0.97, old_data=None, old_loglikelihood=None, tempered_update_prior_weight=0.0,
        log_prob_old_data=0.0, old_cloud=None, savepath=None, particle_store_path=None, loadpath="",
        continue_intermediate=False, save_intermediate=False, intermediate_stage_increment=10, seed=0, device=0,
        max_stages=None, initial_cloud=None, use_graph=0, testing=False, parallel=False, data_vintage="", old_vintage="",
        smc_iteration=1, run_test=False, filestring_addl=(), intermediate_stage_start=0, regime_switching=False, toggle=True,
        debug_assertion=False):
    """Sequential Monte Carlo on one MI355X.  Keyword names follow src/smc_main.jl:119-161 (λ -> lam, n_Φ -> n_phi,
    α -> alpha).  Returns (cloud, w, W) - the three objects the reference writes to `savepath` - and, when `savepath`
    is given, stores them as a numpy .npz (the reference's JLD2/HDF5 writers are outside the hot path).

    Tempered update (src/smc_main.jl:244-333): pass `old_data` (and `old_loglikelihood` if it differs) together with
    `old_cloud`, the Cloud of the previous estimation; the initial cloud is then built on the device from the old cloud
    (same-size continuation, or bridge resample + prior draws when tempered_update_prior_weight > 0 / sizes differ).
    `initial_cloud` instead starts the recursion from a ready-made cloud.

    The remaining reference keywords are accepted for call compatibility: `testing` suppresses the file output like the
    reference's (src/smc_main.jl:513); `parallel` is moot (the device is the parallelism); `data_vintage`, `old_vintage`,
    `smc_iteration`, `run_test`, `filestring_addl`, `intermediate_stage_start`, `toggle`, `debug_assertion` only label or
    guard things that do not exist here.  A tempered update without `old_cloud`
    loads the old cloud from `loadpath` (src/smc_main.jl:245-246).

    Intermediate saves (src/smc_main.jl:499-507): with `save_intermediate`, every `intermediate_stage_increment` stages the
    device loop pauses and {cloud, w, W, j} go to `savepath` with `_stage=<i>` inserted before the extension;
    `continue_intermediate` + `loadpath` resumes from such a file (src/smc_main.jl:334-335, 355-361: stage index, c, ϕ_prop =
    schedule[j] are restored, `resampled_last_period` restarts as false like the reference's).  Files are numpy .npz here;
    `particle_store_path` receives the n_parts x n_para draws (the reference's HDF5 `smcparams`) as .npy."""
    if verbose not in VERBOSITY:
        raise ValueError("verbose must be one of :none, :low, :high")
    if resampling_method not in ("systematic", "multinomial", "polyalgo"):
        raise ValueError("Invalid resampler in SMC. Options are :systematic, :multinomial, or :polyalgo")
    if not 0.0 <= tempered_update_prior_weight <= 1.0:
        raise ValueError("The keyword tempered_update_prior_weight must be within the interval [0, 1]")
    # regime switching (src/smc_main.jl:207-234, src/mutation.jl:98-110): every regime value beyond the first is one more column of the
    # cloud with its own prior / bounds / fixedness; the likelihood closure receives the flattened vector (regime_values() splits
    # it per key).  `toggle` only concerns ModelConstructors' internal regime pointer, which this mirror does not have.
    parameters = flatten_regimes(parameters, regime_switching)
    d = len(parameters)
    if all(p.fixed for p in parameters):
        raise AssertionError("All model parameters are fixed!")
    # (a device family sees the same flattened vector a closure does - regime-1 values of every parameter, then the other regimes'
    # values key by key: its parameter layout must cover all those columns)
    if old_data is not None and np.size(old_data) and initial_cloud is None and old_cloud is None and not continue_intermediate:
        if not loadpath:
            raise ValueError("a tempered update (non-empty old_data) needs old_cloud = the Cloud of the old estimation, or loadpath")
        from .cloudio import load_cloud
        old_cloud = load_cloud(loadpath)[0]                       # cloud_isempty(old_cloud) ? load(loadpath, "cloud") : old_cloud
    data = np.asarray(data, dtype=np.float64)
    device_lik = isinstance(loglikelihood, DeviceLikelihood)
    lik = loglikelihood.spec(data) if device_lik else ("host_callback", [], None, None)
    old_lik = None
    tempered = old_data is not None and np.size(old_data) > 0
    if tempered:
        ol = old_loglikelihood if old_loglikelihood is not None else loglikelihood
        old_lik = ol.spec(np.asarray(old_data, dtype=np.float64)) if isinstance(ol, DeviceLikelihood) else ("host_callback", [], None, None)
    if tempered and (lik[0] == "host_callback") != (old_lik[0] == "host_callback"):
        # one likelihood on the device and the other a host closure: the callback path scores both on the host, the device path both on
        # the device - a mixed pair would silently lose the old likelihood (csrc/callback.hpp); smcmi_run rejects it as well
        raise NotImplementedError("tempered update: loglikelihood and old_loglikelihood must both be DeviceLikelihood objects or both "
                                  "Python callables (wrap the device family in a callable, or pass a DeviceLikelihood for both)")
    if max_stages is None:
        # capacity of the per-stage records and of the two N x max_stages history matrices (16 N max_stages bytes on the device): an
        # adaptive run at the default tempering target takes ~0.9 n_phi stages, so 4 n_phi + 64 is generous; a run that needs
        # more fails with the capacity error and can be repeated with max_stages= (the reference grows its matrices by hcat)
        max_stages = n_phi if use_fixed_schedule else 4 * n_phi + 64
    spec = _spec_from(parameters, lik, old_lik)
    eng = Engine(n_parts, d, seed=seed, device=device, max_stages=max_stages, store_history=True)

    def set_model_on(e, new_lik, new_fn, old_lik_, old_fn, old_dat):
        """priors + the (new, old) likelihood pair on engine e: device families through smcmi_set_likelihood, Python callables
        through smcmi_set_likelihood_callback (the reference's per-particle closure, batched by a trampoline)."""
        e.set_parameters(spec["priors"], spec["bounds"], spec["fixed"])
        if new_lik[0] == "host_callback":
            e.set_likelihood_callback(_batch(new_fn, data), which=0)
        else:
            e.set_likelihood(*new_lik, which=0)
        if old_lik_ is None:
            e.set_likelihood("none", which=1)
        elif old_lik_[0] == "host_callback":
            e.set_likelihood_callback(_batch(old_fn, old_dat), which=1)
        else:
            e.set_likelihood(*old_lik_, which=1)

    old_fn = (old_loglikelihood if old_loglikelihood is not None else loglikelihood) if tempered else None
    old_dat = np.asarray(old_data, dtype=np.float64) if tempered else None
    set_model_on(eng, lik, loglikelihood, old_lik, old_fn, old_dat)
    kw = dict(n_blocks=n_blocks, n_mh_steps=n_mh_steps, lam=lam, n_phi=n_phi, resampling_method=resampling_method,
              threshold_ratio=threshold_ratio, c=c, alpha=alpha, target=target, use_fixed_schedule=use_fixed_schedule,
              tempering_target=tempering_target, prior_weight=tempered_update_prior_weight,
              log_prob_old_data=log_prob_old_data)
    if verbose != "none":
        print("\n\n SMC starts ....\n")
    w0 = None
    if initial_cloud is not None:
        eng.upload_cloud(initial_cloud.particles if isinstance(initial_cloud, Cloud) else initial_cloud)
    elif tempered and old_cloud is not None:
        def prior_engine(n_pr):
            # prior draws scored by the OLD likelihood on the old data (smc_main.jl:288-291)
            pri = Engine(n_pr, d, seed=seed, device=device, max_stages=2, store_history=False)
            pri.set_parameters(spec["priors"], spec["bounds"], spec["fixed"])
            if old_lik[0] == "host_callback":
                pri.set_likelihood_callback(_batch(old_fn, old_dat), which=0)
            else:
                pri.set_likelihood(*old_lik, which=0)
            pri.set_likelihood("none", which=1)
            pri.init_from_prior()
            return pri
        kw["initial_ess"] = _tempered_update_cloud(eng, old_cloud, n_parts, tempered_update_prior_weight, resampling_method, seed,
                                                   device, prior_engine)
        w0 = eng.download_cloud()[:, d + 4].copy()
    else:
        eng.init_from_prior()                             # device prior draws (every prior family); log-likelihoods by the device family or the callback
    cont, elapsed = False, 0.0
    if continue_intermediate:
        cont = _load_intermediate(eng, loadpath, n_phi, lam, d)
        from .cloudio import load_arrays
        elapsed = float(load_arrays(loadpath).get("total_sampling_time", 0.0))
    while True:
        stop = 0
        if save_intermediate:
            i_now = eng.get_loop_state()["stage_index"] if cont else 1
            stop = (i_now // intermediate_stage_increment + 1) * intermediate_stage_increment
        r = eng.run(use_graph=use_graph, stop_after_stage=stop, continue_run=cont, **kw)
        elapsed += r["seconds"]                  # cloud.total_sampling_time accumulates over the stages (smc_main.jl:489-490)
        r["seconds"] = elapsed
        if not r["paused"]:
            break
        _save_intermediate(eng, savepath, r, n_phi)
        cont = True
    rec = eng.stage_records(r["n_stages"])
    w, W = eng.history(r["n_stages"])
    if w0 is not None:                       # W_matrix[:, 1] of a tempered update (smc_main.jl:364-365)
        W = np.array(W)
        W[:, 0] = w0 * n_parts if w0.sum() <= 1.0 else w0
    P = eng.download_cloud()
    cloud = Cloud(d, n_parts)
    cloud.particles = P
    cloud.tempering_schedule = rec["schedule"]
    cloud.ESS = rec["ess"]
    cloud.stage_index = r["n_stages"]
    cloud.n_Phi = n_phi
    cloud.resamples = r["resamples"]
    cloud.c = r["c"]
    cloud.accept = r["accept"]
    cloud.total_sampling_time = r["seconds"]
    cloud.logmdd = r["logmdd"]
    if verbose != "none":
        print(" SMC finished: %d stages, %d resamples, c = %.4f, accept = %.4f, log-MDD = %.6f, %.3f s" %
              (r["n_stages"], r["resamples"], r["c"], r["accept"], r["logmdd"], r["seconds"]))
        if verbose == "high":
            mu, sd = weighted_mean(cloud), weighted_std(cloud)
            for p, m_, s_ in zip(parameters, mu, sd):
                print("   %-12s mean %12.6f  std %12.6f" % (p.key, m_, s_))
    if particle_store_path and not testing:
        from .cloudio import save_smcparams
        save_smcparams(particle_store_path, cloud.particles, d)                            # `smcparams`, smc_main.jl:514-520
    if savepath and not testing:
        from .cloudio import save_cloud
        save_cloud(savepath, cloud, w, W)                          # write(file, "cloud"/"w"/"W"), smc_main.jl:521-525
    eng.close()
    return cloud, w, W


def _stage_path(savepath, stage):
    """replace(savepath, ".jld2" => "_stage=$(cloud.stage_index).jld2") (src/smc_main.jl:500) for any extension."""
    root, ext = os.path.splitext(savepath if savepath else "smc_cloud.npz")
    return "%s_stage=%d%s" % (root, stage, ext or ".npz")


def _save_intermediate(eng, savepath, r, n_phi):
    """{cloud, w, W, j} of a paused run (src/smc_main.jl:499-507)."""
    ls = eng.get_loop_state()
    ns = ls["stage_index"]
    rec = eng.stage_records(ns)
    w, W = eng.history(ns)
    from .cloudio import save_arrays
    save_arrays(_stage_path(savepath, ns), cloud_particles=eng.download_cloud(), cloud_tempering_schedule=rec["schedule"], cloud_ESS=rec["ess"],
                c_hist=rec["c_hist"], accept_hist=rec["accept_hist"], resampled=rec["resampled"], cloud_stage_index=ns, cloud_n_Phi=n_phi,
                cloud_resamples=ls["resamples"], cloud_c=ls["c"], cloud_accept=ls["accept"], cloud_total_sampling_time=r["seconds"], w=w, W=W,
                j=ls["j"], logmdd=ls["logmdd"])


def _load_intermediate(eng, loadpath, n_phi, lam, d):
    """continue_intermediate (src/smc_main.jl:334-335, 355-361): cloud, w, W, j from `loadpath`; i = cloud.stage_index,
    c = cloud.c, ϕ_prop = proposed_fixed_schedule[j]; resampled_last_period is not part of the file and restarts as false."""
    if not loadpath:
        raise ValueError("continue_intermediate needs loadpath")
    from .cloudio import load_arrays
    z = load_arrays(loadpath)
    P = np.asfortranarray(z["particles"], dtype=np.float64)
    if P.shape != (eng.n, d + 5):
        raise ValueError("cloud in %s has shape %r, expected %r" % (loadpath, P.shape, (eng.n, d + 5)))
    ns, j = int(z["stage_index"]), int(z["j"])
    w, W = np.asfortranarray(z["w"]), np.asfortranarray(z["W"])
    sched = (np.arange(n_phi) / (n_phi - 1.0)) ** lam
    # log-MDD so far = Σ_n log((1/N) Σ_i w[i,n] W[i,n-1]) - the reference evaluates it from the matrices at the end
    logmdd = float(z["logmdd"]) if "logmdd" in z else float(np.sum(np.log(np.sum(w[:, 1:ns] * W[:, :ns - 1], axis=0) / eng.n)))
    eng.upload_cloud(P)
    eng.set_stage_records(z["tempering_schedule"], z["ESS"], z["c_hist"], z["accept_hist"], z["resampled"])
    eng.set_history(w[:, :ns], W[:, :ns])
    eng.set_loop_state(stage_index=ns, j=j, resampled_last_period=0, resamples=int(z["resamples"]),
                       phi_n=float(z["tempering_schedule"][ns - 1]), phi_prop=float(sched[j - 1]), c=float(z["c"]),
                       accept=float(z["accept"]), ess=float(z["ESS"][ns - 1]), logmdd=logmdd)
    return True


def _tempered_update_cloud(eng, old_cloud, n_parts, prior_weight, resampling_method, seed, device, prior_engine):
    """Initial cloud of a tempered update, built on the device (src/smc_main.jl:244-333).  Returns cloud.ESS[1].
    RNG contract: bridge resample = stage 0, clean-up resample = stage 1, prior draws = init streams of ids 0.."""
    oldP = np.asfortranarray(old_cloud.particles, dtype=np.float64)
    old_n, d = oldP.shape[0], oldP.shape[1] - 5
    if prior_weight == 0.0 and old_n == n_parts:                           # :249-260
        eng.upload_cloud(oldP)
        eng.initialize_likelihoods()
        return float(np.asarray(old_cloud.ESS)[-1])
    n_to = int(round((1.0 - prior_weight) * n_parts))                       # :262-264
    n_pr = n_parts - n_to
    if n_to > 0:
        old = Engine(old_n, d, seed=seed, device=device, max_stages=2, store_history=False)
        try:
            old.upload_cloud(oldP)
            eng.bridge_resample_from(old, n_to, method=resampling_method, stage=0)     # :266-279
        finally:
            old.close()
    if n_pr > 0:
        pri = prior_engine(n_pr)                                            # old_loglikelihood on old_data, :288-291
        try:
            eng.copy_rows_from(pri, n_pr, dst_row0=n_to)                    # vcat, :296
        finally:
            pri.close()
    eng.initialize_likelihoods()                                            # :308
    eng.normalize_weights(zero_bad_loglh=True)                              # :313-314
    eng.resample(resampling_method, stage=1)                                # :317-322 (incl. reset_weights!)
    return float(n_parts)                                                   # push!(cloud.ESS, n_parts), :325


def _safe_call(f, th, data):
    """The reference maps ParamBoundsError / LAPACK / PosDef / Singular / DomainError inside the likelihood to -Inf
    (src/mutation.jl:112-121); the Python equivalents are ArithmeticError, ValueError and numpy LinAlgError."""
    try:
        v = float(f(th, data))
    except (ArithmeticError, ValueError, np.linalg.LinAlgError):
        return -math.inf
    return -math.inf if math.isnan(v) else v


def _batch(fn, data):
    """Batch form of the reference's closure `loglikelihood(parameters, data)` for smcmi_set_likelihood_callback: one call per
    proposal that passed the bounds check (the engine packs them), exceptions the reference maps to -Inf handled per particle.
    A callable with a true `batched` attribute is handed the whole (m, d) array at once."""
    if getattr(fn, "batched", False):
        return lambda th: fn(th, data)
    return lambda th: np.array([_safe_call(fn, th[k], data) for k in range(th.shape[0])])


# ----------------------------------------------------------------------------------------------- the other exported functions
# (src/SMC.jl:14-17: mutation, resample, mvnormal_mixture_draw, initial_draw!, get_cloud).  They run on the device through
# small one-off engines; inside smc() the same steps are fused kernels of the stage loop.
def get_cloud(filepath):
    """src/util.jl:113-115."""
    from .cloudio import load_cloud
    return load_cloud(filepath)[0]


def resample(weights, n_parts=None, method="systematic", parallel=False, seed=0, stage=0, device=0):
    """src/resample.jl:23-72: ancestor indices (0-based) of `n_parts` slots drawn from `weights`.  RNG: the build's Philox contract
    (systematic: one offset; multinomial: one uniform per slot), keyed by `seed` / `stage`."""
    if method == "polyalgo":
        method = "multinomial"
    if method not in ("systematic", "multinomial"):
        raise ValueError("Invalid resampler in SMC. Options are :systematic, :multinomial, or :polyalgo")
    w = np.ascontiguousarray(weights, dtype=np.float64).ravel()
    n_out = int(n_parts) if n_parts is not None else w.size
    src = Engine(w.size, 1, seed=seed, device=device, max_stages=2, store_history=False)
    dst = src if n_out == w.size else Engine(n_out, 1, seed=seed, device=device, max_stages=2, store_history=False)
    try:
        P = np.zeros((w.size, 6), order="F")
        P[:, 5] = w
        src.upload_cloud(P)
        if dst is src:
            return src.resample(method, stage=stage)
        return dst.bridge_resample_from(src, n_out, method=method, stage=stage)
    finally:
        src.close()
        if dst is not src:
            dst.close()


def _one_particle_engine(parameters, lik, old_lik, seed, device):
    spec = _spec_from(parameters, lik, old_lik)
    eng = Engine(1, len(parameters), seed=seed, device=device, max_stages=2, store_history=False)
    eng.set_model(spec)
    return eng


def mvnormal_mixture_draw(theta_old, mu, Sigma, c=1.0, alpha=1.0, seed=0, pid=0, stage=0, t=0, device=0):
    """src/helpers.jl:87-100 for one parameter block: θ_new from the mixture of N(θ_old, c²Σ), N(θ_old, diag c²Σ) and N(μ, c²Σ) with
    weights α, (1-α)/2, (1-α)/2.  The draw belongs to the stream of particle `pid` at (stage, t) of the RNG contract."""
    th = np.ascontiguousarray(theta_old, dtype=np.float64)
    d = th.size
    pars = [parameter("p%d" % k, 0.0, (-1e300, 1e300), prior=Normal(0.0, 1.0)) for k in range(d)]
    eng = Engine(pid + 1, d, seed=seed, device=device, max_stages=2, store_history=False)
    try:
        eng.set_model(_spec_from(pars, ("gauss_iso", [1.0], np.zeros(d), None), None))
        P = np.zeros((pid + 1, d + 5), order="F")
        P[pid, :d] = th
        eng.upload_cloud(P)
        prop, _, _ = eng.propose(np.asarray(mu, dtype=np.float64), np.asarray(Sigma, dtype=np.float64), [0, d], np.arange(d), 0, t, c, alpha, stage)
        return prop[pid].copy()
    finally:
        eng.close()


def mutation(loglikelihood, parameters, data, p, d_mu, d_Sigma, n_free_para, blocks_free, blocks_all, phi_n, phi_n1, c=1.0, alpha=1.0,
             n_mh_steps=1, old_data=None, old_loglikelihood=None, seed=0, pid=0, stage=2, device=0):
    """src/mutation.jl:56-138 for ONE particle row `p` (length n_para + 5): n_mh_steps x blocks of mixture random-walk MH moves at
    ϕ_n.  `blocks_free` / `blocks_all` are the reference's lists of index lists (0-based here).  Device likelihoods only.
    Returns the mutated row; the acceptance column holds Σ accepted block lengths / n_free (quirk Q2)."""
    if not isinstance(loglikelihood, DeviceLikelihood):
        raise NotImplementedError("mutation() takes a device likelihood; host callbacks run through smc()")
    parameters = list(parameters)
    d = len(parameters)
    row = np.ascontiguousarray(p, dtype=np.float64)
    lik = loglikelihood.spec(np.asarray(data, dtype=np.float64))
    old_lik = None
    if old_data is not None and np.size(old_data):
        ol = old_loglikelihood if old_loglikelihood is not None else loglikelihood
        old_lik = ol.spec(np.asarray(old_data, dtype=np.float64))
    eng = Engine(pid + 1, d, seed=seed, device=device, max_stages=2, store_history=False)
    try:
        eng.set_model(_spec_from(parameters, lik, old_lik))
        P = np.zeros((pid + 1, d + 5), order="F")
        P[pid] = row
        eng.upload_cloud(P)
        bf = np.concatenate([np.asarray(b, dtype=np.int32) for b in blocks_free])
        bp = np.concatenate([[0], np.cumsum([len(b) for b in blocks_free])]).astype(np.int32)
        eng.mutate(np.asarray(d_mu, dtype=np.float64), np.asarray(d_Sigma, dtype=np.float64), bp, bf, phi_n, phi_n1, c, alpha, n_mh_steps, stage)
        return eng.download_cloud()[pid].copy()
    finally:
        eng.close()


def initial_draw(loglikelihood, parameters, data, cloud, parallel=False, regime_switching=False, toggle=True, seed=0, device=0):
    """initial_draw! (src/initialization.jl:88-119): fills `cloud` (n_parts x n_para + 5) with prior draws whose log-likelihood is
    finite, their loglh / logprior, old_loglh = 0, weight 1.  Device likelihoods only; returns the cloud."""
    if not isinstance(loglikelihood, DeviceLikelihood):
        raise NotImplementedError("initial_draw() takes a device likelihood; host callbacks run through smc()")
    parameters = flatten_regimes(list(parameters), regime_switching)      # (the regime columns are drawn like any other, smc_main.jl:207-234)
    n, d = len(cloud), len(parameters)
    eng = Engine(n, d, seed=seed, device=device, max_stages=2, store_history=False)
    try:
        eng.set_model(_spec_from(parameters, loglikelihood.spec(np.asarray(data, dtype=np.float64)), None))
        eng.init_from_prior()
        cloud.particles = eng.download_cloud()
    finally:
        eng.close()
    return cloud
