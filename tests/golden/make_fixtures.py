#!/opt/conda/bin/python3.9
"""Extract golden vectors (DATA only) from the reference's own test fixtures.

Run in the build container (the reference tree is not available on the GPU box):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_fixtures.py

Needs h5py (conda python 3.9 has it; the system python3.10 does not).  JLD2 files are
HDF5; Julia arrays appear with reversed dims, so every matrix is transposed back to the
Julia shape (rows = particles).  Outputs small .npz files next to this script.  Nothing
but numeric inputs / expected outputs of the reference's tests is written - no source.

Sources (all under /root/reference):
  test/reference/ess_inputs_version=150.jld2, ess_output_version=150.jld2   (test/helpers.jl:133-175)
  test/reference/solve_adaptive_phi.jld2, helpers_output_version=150.jld2   (test/helpers.jl:15-53)
  test/reference/proposal_densities_in.jld2, proposal_densities_output_*    (test/helpers.jl:101-127)
  test/reference/mvnormal_inputs.jld2                                       (test/helpers.jl:58-80)
  test/reference/mutation_inputs.jld2, mutation_outputs_version=150.jld2    (test/mutation.jl:22-59)
  test/reference/initial_draw_out_*, initialize_likelihood_out_*, one_draw_out_*, draw_likelihood_out_*,
  test/reference/test_data.h5                                               (test/initialization.jl, test/modelsetup.jl)
  test/save/output_data/an_schorfheide/ss0/estimate/raw/smc_cloud_npart=1000_vint=000000.jld2
  examples/regression_model/save/input_data/reg_data.jld2, examples/data/capm.jld2
"""
import os
import h5py
import numpy as np

REF = "/root/reference"
R = REF + "/test/reference/"
OUT = os.path.dirname(os.path.abspath(__file__))


def jl(a):
    return np.ascontiguousarray(np.asarray(a).T)


def cloud(f, key):
    c = f[key][()]
    d = {}
    for n in c.dtype.names:
        v = c[n]
        d[n] = f[v][()] if isinstance(v, h5py.Reference) else v
    d["particles"] = jl(d["particles"])
    return d


def mvn(f, key):
    d = f[key][()]
    mu = f[d["μ"]][()]
    Sigma = jl(f[d["Σ"]["mat"]][()])
    return np.asarray(mu, dtype=np.float64), np.asarray(Sigma, dtype=np.float64)


def jbool(x):
    return bool(np.frombuffer(np.asarray(x).tobytes(), dtype=np.uint8)[0])


def save(name, **kw):
    p = os.path.join(OUT, name)
    np.savez_compressed(p, **kw)
    print("%-28s %8d bytes" % (name, os.path.getsize(p)))


# 1. compute_ESS
fi = h5py.File(R + "ess_inputs_version=150.jld2", "r")
fo = h5py.File(R + "ess_output_version=150.jld2", "r")
save("ess.npz", loglh=fi["loglh"][()], weights=fi["current_weights"][()], old_loglh=fi["old_loglh"][()],
     phi_n=fi["ϕ_n"][()], phi_n1=fi["ϕ_n1"][()], ess=fo["ess"][()])

# 2. solve_adaptive_phi
f = h5py.File(R + "solve_adaptive_phi.jld2", "r")
c = cloud(f, "cloud")
fo = h5py.File(R + "helpers_output_version=150.jld2", "r")
save("adaptive_phi.npz", particles=c["particles"], cloud_ess=np.asarray(c["ESS"], dtype=np.float64),
     i=f["i"][()], j=f["j"][()], phi_n1=f["phi_n1"][()], phi_prop=f["phi_prop"][()],
     schedule=f["proposed_fixed_schedule"][()], target=f["tempering_target"][()],
     resampled_last=jbool(f["resampled_last_period"][()]),
     out_phi_n=fo["phi_n"][()], out_j=fo["j"][()], out_phi_prop=fo["phi_prop"][()],
     out_resampled_last=jbool(fo["resampled_last_period"][()]))

# 3. compute_proposal_densities (+ mvnormal_mixture_draw inputs)
f = h5py.File(R + "proposal_densities_in.jld2", "r")
mu, Sig = mvn(f, "d_subset")
fo = h5py.File(R + "proposal_densities_output_version=150.jld2", "r")
save("proposal_densities.npz", mu=mu, Sigma=Sig, c=f["c"][()], alpha=f["α"][()],
     para_draw=f["para_draw"][()], para_subset=f["para_subset"][()], q0=fo["q0"][()], q1=fo["q1"][()])
f = h5py.File(R + "mvnormal_inputs.jld2", "r")
mu, Sig = mvn(f, "d_subset")
save("mvnormal_inputs.npz", mu=mu, Sigma=Sig, c=f["c"][()], alpha=f["α"][()], para_subset=f["para_subset"][()])

# 4. linear test model: data + (theta, loglh, logprior) triples
t = h5py.File(R + "test_data.h5", "r")
data, X = jl(t["data"][()]), jl(t["X"][()])
f1 = h5py.File(R + "initial_draw_out_version=150.jld2", "r")
f2 = h5py.File(R + "initialize_likelihood_out_version=150.jld2", "r")
c1, c2 = cloud(f1, "cloud"), cloud(f2, "init_lik_cloud")
f3 = h5py.File(R + "one_draw_out_version=150.jld2", "r")
od = [np.asarray(f3[r][()]).ravel() for r in f3["draw"][()]]
f4 = h5py.File(R + "draw_likelihood_out_version=150.jld2", "r")
dl = [np.asarray(f4[r][()]).ravel() for r in f4["draw_lik"][()]]
save("linmodel.npz", data=data, X=X, initial_draw=c1["particles"], init_lik=c2["particles"],
     one_draw_theta=od[0], one_draw_loglh=od[1], one_draw_logprior=od[2],
     draw_lik_loglh=dl[0], draw_lik_logprior=dl[1])

# 4b. the regime-switching version of the linear test model (test/regime_switching_smc.jl, test/modelsetup.jl:78-103): data and predictors
save("rsmodel.npz", rsdata=jl(t["rsdata"][()]), Xrs=jl(t["Xrs"][()]))

# 5. mutation (reject-path identity + stored logprior)
f = h5py.File(R + "mutation_inputs.jld2", "r")
mu, Sig = mvn(f, "d")
pin = cloud(f, "particles")
bf = [np.asarray(f[r][()]) for r in f["blocks_free"][()]]
ba = [np.asarray(f[r][()]) for r in f["blocks_all"][()]]
fo = h5py.File(R + "mutation_outputs_version=150.jld2", "r")
pout = cloud(fo, "particles")
save("mutation.npz", particles_in=pin["particles"], particles_out=pout["particles"], mu=mu, Sigma=Sig,
     blocks_free=np.concatenate(bf), blocks_all=np.concatenate(ba), block_sizes=np.array([len(b) for b in bf]),
     c=f["c"][()], alpha=f["α"][()], phi_n=f["ϕ_n"][()], phi_n1=f["ϕ_n1"][()], old_data=jl(f["old_data"][()]))

# 6. 99-stage replay of correction / ESS / resample bookkeeping + implied log-MDD
p = REF + "/test/save/output_data/an_schorfheide/ss0/estimate/raw/smc_cloud_npart=1000_vint=000000.jld2"
f = h5py.File(p, "r")
c = cloud(f, "cloud")
w, W = jl(f["w"][()]), jl(f["W"][()])
N, S = w.shape
logmdd = float(np.sum(np.log(np.sum(w[:, 1:] * W[:, :-1], axis=0) / N)))
save("replay_as1000.npz", w=w, W=W, ess=np.asarray(c["ESS"], dtype=np.float64),
     schedule=np.asarray(c["tempering_schedule"], dtype=np.float64), resamples=int(c["resamples"]),
     accept=float(c["accept"]), accept_col=c["particles"][:, -2], logmdd=logmdd,
     total_sampling_time=float(c["total_sampling_time"]))
print("replay logMDD", repr(logmdd), "resamples", int(c["resamples"]))

# 7. example data sets (config inputs)
f = h5py.File(REF + "/examples/regression_model/save/input_data/reg_data.jld2", "r")
save("reg_data.npz", data=jl(f["data"][()]))
f = h5py.File(REF + "/examples/data/capm.jld2", "r")
save("capm_data.npz", lik_data=jl(f["lik_data"][()]), market_data=jl(f["market_data"][()]))
