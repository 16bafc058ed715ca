# SMCMI.jl - Julia `ccall` shim over libsmcmi.so (include/smcmi.h).
#
# Keeps the reference entry point `smc(loglikelihood, parameters, data; kwargs...)` (src/smc_main.jl:118-161) and hands the
# correction / selection / mutation loop to the MI355X engine.  Julia is not available in the build image, so this file has
# never been executed; it only marshals arguments (a `Cloud`'s `particles` matrix is already the column-major N x (d+5)
# buffer the C ABI expects, so uploads/downloads are zero-conversion).  See INTEGRATION.md.
module SMCMI

using ModelConstructors, Distributions

const LIB = get(ENV, "SMCMI_LIB", joinpath(@__DIR__, "..", "csrc", "libsmcmi.so"))

struct Config
    n_parts::Int64; n_local::Int64; gid0::Int64; n_para::Int32; device::Int32
    seed::UInt64; max_stages::Int32; store_history::Int32
end
struct RunConfig
    n_blocks::Int32; n_mh_steps::Int32; lambda::Float64; n_phi::Int32; resampling_method::Int32
    threshold_ratio::Float64; c::Float64; alpha::Float64; target::Float64; use_fixed_schedule::Int32
    tempering_target::Float64; tempered_update_prior_weight::Float64; log_prob_old_data::Float64
    solver_passes::Int32; sync_every::Int32; use_graph::Int32; initial_ess::Float64; phi_rtol::Float64
    stop_after_stage::Int32; continue_run::Int32   # save_intermediate / continue_intermediate (smc_main.jl:334-361, 499-507)
end
mutable struct Result
    n_stages::Int32; resamples::Int32; logmdd::Float64; c::Float64; accept::Float64; seconds::Float64
    kernel_ms_mutate::Float64; n_mutate_launches::Int32; solver_passes::Int64; solver_stalls::Int32; select_stalls::Int32; spec_stalls::Int32; paused::Int32
    Result() = new(0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0)
end

check(rc) = rc == 0 ? nothing : error("smcmi error $rc: " * unsafe_string(ccall((:smcmi_last_error, LIB), Cstring, ())))

# device likelihood families standing in for the user closure (SMCMI_LIK_* in smcmi.h)
struct GaussIso; sigma::Float64; end
struct LinReg; sigma2::Float64; end
struct LinModel3; X::Matrix{Float64}; end
struct CapmLiteral; market::Matrix{Float64}; end
const DeviceLikelihood = Union{GaussIso, LinReg, LinModel3, CapmLiteral}
family(::GaussIso) = Int32(0); family(::LinReg) = Int32(1); family(::LinModel3) = Int32(2); family(::CapmLiteral) = Int32(3)
lik_par(l::GaussIso) = [l.sigma]; lik_par(l::LinReg) = [l.sigma2]; lik_par(::Any) = Float64[]
lik_aux(l::LinModel3) = l.X; lik_aux(l::CapmLiteral) = l.market; lik_aux(::Any) = zeros(0, 0)

prior_code(d::Normal) = (Int32(0), d.μ, d.σ)
prior_code(d::Uniform) = (Int32(1), d.a, d.b)
prior_code(d::Gamma) = (Int32(2), shape(d), scale(d))
prior_code(d::Beta) = (Int32(3), d.α, d.β)
prior_code(d::InverseGamma) = (Int32(4), shape(d), scale(d))
prior_code(d::ModelConstructors.RootInverseGamma) = (Int32(5), d.ν, d.τ)

const RESAMPLER = Dict(:systematic => Int32(0), :multinomial => Int32(1), :polyalgo => Int32(1))

"""
    smc(loglikelihood, parameters, data; kwargs...) -> (particles, w, W, result)

Same keyword arguments as SMC.smc.  `loglikelihood` is a `DeviceLikelihood`; arbitrary Julia closures go through
`smcmi_propose` / `smcmi_accept` (see `smc_callback` below).  The caller wraps `particles` into `SMC.Cloud` and saves
`{cloud, w, W}` exactly as src/smc_main.jl:513-526 does.
"""
function smc(loglikelihood::DeviceLikelihood, parameters::ParameterVector, data::Matrix{Float64};
             n_parts::Int = 5_000, n_blocks::Int = 1, n_mh_steps::Int = 1, λ::Float64 = 2.1, n_Φ::Int = 300,
             resampling_method::Symbol = :systematic, threshold_ratio::Float64 = 0.5, c::Float64 = 0.5, α::Float64 = 1.0,
             target::Float64 = 0.25, use_fixed_schedule::Bool = true, tempering_target::Float64 = 0.97,
             tempered_update_prior_weight::Float64 = 0.0, log_prob_old_data::Float64 = 0.0, seed::Integer = 0,
             device::Integer = 0, initial_cloud::Union{Nothing, Matrix{Float64}} = nothing)
    haskey(RESAMPLER, resampling_method) || throw("Invalid resampler in SMC. Options are :systematic, :multinomial, or :polyalgo")
    d = length(parameters)
    max_stages = use_fixed_schedule ? n_Φ : 20 * n_Φ
    h = Ref{Ptr{Cvoid}}(C_NULL)
    cfg = Config(n_parts, n_parts, 0, d, device, seed, max_stages, 1)
    check(ccall((:smcmi_create, LIB), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, h))
    try
        fixed = Int32[p.fixed ? 1 : 0 for p in parameters]
        lo = Float64[p.valuebounds[1] for p in parameters]; hi = Float64[p.valuebounds[2] for p in parameters]
        codes = [p.fixed ? (Int32(0), p.value, 1.0) : prior_code(p.prior.value) for p in parameters]
        fam = Int32[x[1] for x in codes]; pa = Float64[x[2] for x in codes]; pb = Float64[x[3] for x in codes]
        check(ccall((:smcmi_set_parameters, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
                    h[], fixed, lo, hi, fam, pa, pb))
        par = lik_par(loglikelihood); aux = lik_aux(loglikelihood)
        check(ccall((:smcmi_set_likelihood, LIB), Cint,
                    (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64),
                    h[], 0, family(loglikelihood), par, length(par), data, size(data, 1), size(data, 2), aux, size(aux, 1), size(aux, 2)))
        check(ccall((:smcmi_set_likelihood, LIB), Cint,
                    (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64),
                    h[], 1, -1, C_NULL, 0, C_NULL, 0, 0, C_NULL, 0, 0))
        if initial_cloud === nothing
            check(ccall((:smcmi_init_from_prior, LIB), Cint, (Ptr{Cvoid},), h[]))      # initial_draw!
        else
            check(ccall((:smcmi_upload_cloud, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], initial_cloud))
        end
        rc = RunConfig(n_blocks, n_mh_steps, λ, n_Φ, RESAMPLER[resampling_method], threshold_ratio, c, α, target,
                       use_fixed_schedule ? 1 : 0, tempering_target, tempered_update_prior_weight, log_prob_old_data, 0, 0, 0, 0.0, 0.0, 0, 0)
        res = Result()
        check(ccall((:smcmi_run, LIB), Cint, (Ptr{Cvoid}, Ref{RunConfig}, Ref{Result}), h[], rc, res))
        particles = Matrix{Float64}(undef, n_parts, d + 5)
        check(ccall((:smcmi_download_cloud, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], particles))
        w = Matrix{Float64}(undef, n_parts, res.n_stages); W = similar(w)
        check(ccall((:smcmi_get_history, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], w, W))
        return particles, w, W, res
    finally
        ccall((:smcmi_destroy, LIB), Cint, (Ptr{Cvoid},), h[])
    end
end

end # module
