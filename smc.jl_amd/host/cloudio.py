"""Cloud files and the cloud utilities around an estimation (host side; nothing here touches the device):

  save_cloud / load_cloud       the {cloud, w, W} triple the reference writes to `savepath` (src/smc_main.jl:521-525)
  split_cloud / join_cloud      src/particle.jl:542-648 (a large cloud file in `n_pieces` part files and back)
  add_parameters_to_cloud       src/particle.jl:705-760 (bridge to a model that extends the old one by new parameters)

Files are numpy .npz: JLD2 is a Julia-side format (the Julia shim keeps writing it itself); the keys are the
reference's field names (λ / Φ spelt in ASCII), so `load_cloud` reads what `smc(..., savepath=...)` wrote.
"""
import math
import os

import numpy as np

from . import hostmath as hm

_FIELDS = ("tempering_schedule", "ESS", "stage_index", "n_Phi", "resamples", "c", "accept", "total_sampling_time")


def _cloud_cls():
    from .api import Cloud          # api imports this module's functions lazily as well
    return Cloud


_H5_EXT = (".h5", ".hdf5", ".jld2")


def save_arrays(path, **arrays):
    """One file of named arrays at EXACTLY `path`: HDF5 (h5min, Julia array convention) for .h5 / .hdf5 / .jld2 names - the
    reference's `savepath` convention, readable by HDF5.jl / h5py; `cloud_*` keys form the group `cloud` - numpy .npz otherwise
    (written through an open handle: np.savez(path) would append ".npz" to a foreign extension)."""
    if str(path).lower().endswith(_H5_EXT):
        from . import h5min
        items, grp = {}, {}
        for k, v in arrays.items():
            (grp if k.startswith("cloud_") else items)[k[6:] if k.startswith("cloud_") else k] = np.asarray(v)
        if "n_Phi" in grp:
            grp["n_\u03a6"] = grp.pop("n_Phi")          # the reference's field name (src/particle.jl:37: n_Φ), UTF-8 in the file
        if grp:
            items["cloud"] = grp
        h5min.write_julia(path, items)
    else:
        with open(path, "wb") as f:
            np.savez(f, **{(k[6:] if k.startswith("cloud_") else k): v for k, v in arrays.items()})


def load_arrays(path):
    """-> dict of arrays, from either format of save_arrays (sniffed from the file's first bytes)."""
    with open(path, "rb") as f:
        head = f.read(8)
    if head == b"\x89HDF\r\n\x1a\n":
        from . import h5min
        z = h5min.read(path, julia=True)
        out = {k: v for k, v in z.items() if not isinstance(v, dict)}
        out.update(z.get("cloud", {}))
        if "n_\u03a6" in out:
            out["n_Phi"] = out.pop("n_\u03a6")
        return out
    z = np.load(path)
    return {k: z[k] for k in z.files}


def save_cloud(path, cloud, w, W, **extra):
    """write(file, "cloud", cloud); write(file, "w", w); write(file, "W", W) (src/smc_main.jl:521-525).  In an HDF5 file the Cloud's
    fields are the datasets of the group `cloud` (particles as an N x (n_para + 5) Julia matrix); true JLD2 struct encoding is the
    Julia shim's job (smc.jl_amd/julia/SMCMI.jl writes through JLD2.jl itself)."""
    save_arrays(path, cloud_particles=np.asarray(cloud.particles), w=np.asarray(w), W=np.asarray(W),
                **{"cloud_" + k: getattr(cloud, k) for k in _FIELDS}, **extra)


def save_smcparams(path, particles, n_para):
    """`particle_store_path`: the n_parts x n_para draws as HDF5 dataset "smcparams" (src/smc_main.jl:514-520) for .h5 names,
    a .npy array otherwise."""
    P = np.ascontiguousarray(np.asarray(particles)[:, :n_para])
    if str(path).lower().endswith(_H5_EXT):
        from . import h5min
        h5min.write_julia(path, {"smcparams": P})
    else:
        with open(path, "wb") as f:
            np.save(f, P)


def load_cloud(path):
    """-> (cloud, w, W) as `load(path, "cloud")`, `load(path, "w")`, `load(path, "W")`."""
    z = load_arrays(path)
    P = np.asfortranarray(z["particles"], dtype=np.float64)
    cloud = _cloud_cls()(P.shape[1] - 5, P.shape[0])
    cloud.particles = P
    cloud.tempering_schedule, cloud.ESS = np.array(z["tempering_schedule"]), np.array(z["ESS"])
    cloud.stage_index, cloud.n_Phi, cloud.resamples = int(z["stage_index"]), int(z["n_Phi"]), int(z["resamples"])
    cloud.c, cloud.accept, cloud.total_sampling_time = float(z["c"]), float(z["accept"]), float(z["total_sampling_time"])
    return cloud, np.array(z["w"]), np.array(z["W"])


def _part_path(filename, i):
    """replace(filename, ".jld2" => "_part$(i).jld2") (src/particle.jl:574) for any extension."""
    root, ext = os.path.splitext(filename)
    return "%s_part%d%s" % (root, i, ext)


def split_cloud(filename, n_pieces):
    """src/particle.jl:542-582: rows ((i-1) n/n_pieces+1 : i n/n_pieces) of particles, w, W go to `<name>_part<i>`; every part
    carries the whole-cloud scalars and paths (ESS, c, stage_index, total_sampling_time, accept, n_Φ, resamples, schedule)."""
    cloud, w, W = load_cloud(filename)
    n_part = cloud.particles.shape[0]
    if n_part % n_pieces != 0:
        raise AssertionError("mod(n_part, n_pieces) == 0")
    small = n_part // n_pieces
    Cloud = _cloud_cls()
    for i in range(1, n_pieces + 1):
        inds = slice((i - 1) * small, i * small)
        part = Cloud(cloud.particles.shape[1] - 5, small)
        part.particles = np.asfortranarray(cloud.particles[inds, :])
        for k in _FIELDS:
            setattr(part, k, getattr(cloud, k))
        save_cloud(_part_path(filename, i), part, w[inds, :], W[inds, :])


def join_cloud(filename, n_pieces, save_cloud_file=True):
    """src/particle.jl:596-648: vcat of the part files' particles, w, W; scalars and paths from the first part.  Returns
    (cloud, w, W) and, with `save_cloud_file`, writes them to `filename`."""
    parts = [load_cloud(_part_path(filename, i)) for i in range(1, n_pieces + 1)]
    P = np.asfortranarray(np.vstack([c.particles for c, _, _ in parts]))
    w = np.vstack([w_ for _, w_, _ in parts])
    W = np.vstack([W_ for _, _, W_ in parts])
    cloud = _cloud_cls()(P.shape[1] - 5, P.shape[0])
    cloud.particles = P
    for k in _FIELDS:
        setattr(cloud, k, getattr(parts[0][0], k))
    if save_cloud_file:
        save_cloud(filename, cloud, w, W)
    return cloud, w, W


# ---- priors on the host: densities as Distributions.jl / ModelConstructors define them, draws for Normal / Uniform on the build's
# ---- counter-based RNG contract and numpy's Philox for the other families
def prior_logpdf(prior, x):
    fam, a, b = prior.triple()
    if fam == "normal":
        z = (x - a) / b
        return -(z * z + math.log(2.0 * math.pi)) / 2.0 - math.log(b)
    if fam == "uniform":
        return -math.log(b - a) if a <= x <= b else -math.inf
    if fam == "gamma":                      # shape a, scale b
        return -math.inf if x < 0 else -math.lgamma(a) - a * math.log(b) + (a - 1.0) * math.log(x) - x / b
    if fam == "beta":
        return -math.inf if (x < 0 or x > 1) else ((a - 1.0) * math.log(x) + (b - 1.0) * math.log1p(-x) -
                                                   (math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b)))
    if fam == "invgamma":                   # shape a, scale b
        return -math.inf if x <= 0 else a * math.log(b) - math.lgamma(a) - (a + 1.0) * math.log(x) - b / x
    if fam == "rootinvgamma":               # ν = a, τ = b
        return -math.inf if x <= 0 else (math.log(2.0) - math.lgamma(a / 2.0) + (a / 2.0) * math.log(a * b * b / 2.0) -
                                         ((a + 1.0) / 2.0) * math.log(x * x) - a * b * b / (2.0 * x * x))
    raise ValueError("unknown prior family %r" % (fam,))


def logprior(parameters, theta):
    """ModelConstructors.prior(parameters): Σ logpdf over the free parameters."""
    return sum(prior_logpdf(p.prior, float(theta[k])) for k, p in enumerate(parameters) if not p.fixed)


def _prior_draw(p, seed, pid, k, gen):
    """One draw of parameter k for particle pid, rejected until strictly inside valuebounds (src/initialization.jl:23-63)."""
    fam, a, b = p.prior.triple()
    lo, hi = p.valuebounds
    for r in range(100000):
        if fam in ("normal", "uniform"):
            ua, ub = hm.uniform_pair(seed, pid, 0, hm.rng_tag(hm.P_INIT, r, k))
            x = a + b * (math.sqrt(-2.0 * math.log(ua)) * math.cos(2.0 * math.pi * ub)) if fam == "normal" else a + (b - a) * ua
        elif fam == "gamma":
            x = gen.gamma(a, b)
        elif fam == "beta":
            x = gen.beta(a, b)
        elif fam == "invgamma":
            x = b / gen.gamma(a, 1.0)
        elif fam == "rootinvgamma":         # σ with ν τ² / σ² ~ χ²(ν)
            x = math.sqrt(a * b * b / gen.chisquare(a))
        else:
            raise ValueError("unknown prior family %r" % (fam,))
        if lo < x < hi:
            return x
    raise RuntimeError("no prior draw of %s inside its bounds" % p.key)


def add_parameters_to_cloud(old_cloud, parameters, old_para_inds, seed=0):
    """src/particle.jl:705-760 (regime_switching = false): the cloud of an old estimation extended by prior draws of the new
    parameters.  `parameters` is the new model's vector, `old_para_inds` the boolean mask of those the old model had (in the
    old order).  Columns of the result: [old draws scattered into their places | prior draws elsewhere | loglh of the old
    model | logprior of the full vector | old_loglh = 0 | accept, weight of the old cloud]; ESS path of the old cloud,
    stage_index = 1, c = 0, accept = 0.25 - the constructor call at :760."""
    if isinstance(old_cloud, str):
        old_cloud = load_cloud(old_cloud)[0]
    parameters = list(parameters)
    mask = np.asarray(old_para_inds, dtype=bool)
    oldP = np.asarray(old_cloud.particles)
    n_parts, d_old, d = oldP.shape[0], oldP.shape[1] - 5, len(parameters)
    if mask.size != d or int(mask.sum()) != d_old:
        raise ValueError("old_para_inds must flag exactly the %d old parameters among the %d new ones" % (d_old, d))
    gen = np.random.Generator(np.random.Philox(key=int(seed)))
    vals = np.empty((n_parts, d))
    for i in range(n_parts):
        for k, p in enumerate(parameters):
            if mask[k]:
                continue
            vals[i, k] = p.value if p.fixed else _prior_draw(p, seed, i, k, gen)
    vals[:, mask] = oldP[:, :d_old]
    meta = np.empty((n_parts, 5))
    meta[:, 0] = oldP[:, d_old]                                       # loglh of the old model
    meta[:, 1] = [logprior(parameters, vals[i]) for i in range(n_parts)]
    meta[:, 2] = 0.0                                                  # old_loglh: a "new" cloud
    meta[:, 3] = oldP[:, d_old + 3]
    meta[:, 4] = oldP[:, d_old + 4]
    cloud = _cloud_cls()(d, n_parts)
    cloud.particles = np.asfortranarray(np.hstack([vals, meta]))
    cloud.tempering_schedule = np.zeros(1)
    cloud.ESS = np.array(old_cloud.ESS, dtype=np.float64)
    cloud.stage_index, cloud.n_Phi, cloud.resamples = 1, 0, 0
    cloud.c, cloud.accept, cloud.total_sampling_time = 0.0, 0.25, 0.0
    return cloud
