# SMCMI.jl - the reference-side binding: Julia `ccall` shim over libsmcmi.so (include/smcmi.h).
#
# Keeps the reference's entry point and types for the correction / selection / mutation loop:
#
#     smc(loglikelihood::Function, parameters::ParameterVector, data::Matrix; kwargs...)      src/smc_main.jl:118-161
#     Cloud, get_vals / get_loglh / ... / weighted_mean / weighted_cov                          src/particle.jl
#
# and hands the loop (src/smc_main.jl:377-508) to the MI355X engine.  The user's closure is bound through
# `smcmi_set_likelihood_callback`: one `@cfunction` trampoline evaluates the closure on the batch of proposals that passed the
# bounds check, synchronously on the calling thread (the library never calls back from another thread).  `DeviceLikelihood`
# structs select a built-in device family instead (no PCIe traffic inside the loop).
#
# Bound: smc, Cloud and its accessors, and the other exports of src/SMC.jl:13-17 (mutation, resample, mvnormal_mixture_draw,
# initial_draw!, get_cloud, isempty).  `run_test` stops at stage 3 like smc_main.jl:495; `verbose` prints the reference's stage lines
# from the per-stage records after the run; `old_cloud` may be a Cloud of this module or one SMC.jl itself saved.
#
# Julia is not part of the build image, so this file has not been executed here; every `ccall` spells out the C signature of
# include/smcmi.h and the structs mirror its layout field by field (tests/test_abi_cpu.py pins the C side).  See INTEGRATION.md.
module SMCMI

using ModelConstructors, Distributions, Random, Dates
import JLD2, HDF5, LinearAlgebra

export smc, Cloud, get_vals, get_loglh, get_logprior, get_old_loglh, get_logpost, get_accept, get_weights, weighted_mean, weighted_cov,
       weighted_std, cloud_isempty, GaussIso, LinReg, LinModel3, CapmLiteral, LGSSKalman,
       mutation, resample, mvnormal_mixture_draw, initial_draw!, get_cloud                       # src/SMC.jl:13-17

const LIB = get(ENV, "SMCMI_LIB", joinpath(@__DIR__, "..", "csrc", "libsmcmi.so"))
const Handle = Ptr{Cvoid}

# ---- struct mirrors of include/smcmi.h (field order and widths are the ABI)
struct Config                 # smcmi_config
    n_parts::Int64; n_local::Int64; gid0::Int64; n_para::Int32; device::Int32
    seed::UInt64; max_stages::Int32; store_history::Int32
end
struct RunConfig              # smcmi_run_config
    n_blocks::Int32; n_mh_steps::Int32; lambda::Float64; n_phi::Int32; resampling_method::Int32
    threshold_ratio::Float64; c::Float64; alpha::Float64; target::Float64; use_fixed_schedule::Int32
    tempering_target::Float64; tempered_update_prior_weight::Float64; log_prob_old_data::Float64
    solver_passes::Int32; sync_every::Int32; use_graph::Int32; initial_ess::Float64; phi_rtol::Float64
    stop_after_stage::Int32; continue_run::Int32
end
mutable struct Result         # smcmi_result
    n_stages::Int32; resamples::Int32; logmdd::Float64; c::Float64; accept::Float64; seconds::Float64
    kernel_ms_mutate::Float64; n_mutate_launches::Int32; solver_passes::Int64; solver_stalls::Int32; select_stalls::Int32
    spec_stalls::Int32; paused::Int32
    n_segments::Int32; segment_stages::Int32; kernel_ms_segments::Float64
    segment_blocks::Int32; segment_state::Int32; segment_timeouts::Int32; shift_fallback_stage::Int32
    Result() = new(0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0)
end
mutable struct LoopState      # smcmi_loop_state
    stage_index::Int32; j::Int32; resampled_last_period::Int32; resamples::Int32
    phi_n::Float64; phi_prop::Float64; c::Float64; accept::Float64; ess::Float64; logmdd::Float64
    LoopState() = new(0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
end

check(rc) = rc == 0 ? nothing : error("smcmi error $rc: " * unsafe_string(ccall((:smcmi_last_error, LIB), Cstring, ())))

# ---- Cloud: the reference's container (src/particle.jl:31-63), same fields, same column map
mutable struct Cloud
    particles::Matrix{Float64}            # N x (n_para + 5): params | loglh | logprior | old_loglh | accept | weight
    tempering_schedule::Vector{Float64}
    ESS::Vector{Float64}
    stage_index::Int
    n_Φ::Int
    resamples::Int
    c::Float64
    accept::Float64
    total_sampling_time::Float64
end
Cloud(n_params::Int, n_parts::Int) = Cloud(Matrix{Float64}(undef, n_parts, n_params + 5), [0.], [0.], 1, 0, 0, 0., 0.25, 0.)
cloud_isempty(c::Cloud) = isempty(c.particles)
Base.isempty(c::Cloud) = isempty(c.particles)                                                   # src/particle.jl (the reference's own spelling)
Base.length(c::Cloud) = size(c.particles, 1)
# A cloud saved by SMC.jl itself (`load(path, "cloud")::SMC.Cloud`, smc_main.jl:521-525) or by this module: the same nine fields
as_cloud(c::Cloud) = c
as_cloud(c) = Cloud(Matrix{Float64}(getfield(c, :particles)), Vector{Float64}(getfield(c, :tempering_schedule)), Vector{Float64}(getfield(c, :ESS)),
                    Int(getfield(c, :stage_index)), Int(getfield(c, :n_Φ)), Int(getfield(c, :resamples)), Float64(getfield(c, :c)),
                    Float64(getfield(c, :accept)), Float64(getfield(c, :total_sampling_time)))
get_cloud(filepath::String) = as_cloud(JLD2.load(filepath, "cloud"))                             # src/util.jl:113-115
n_para(c::Cloud) = size(c.particles, 2) - 5
get_vals(c::Cloud; transpose::Bool = true) = transpose ? collect(c.particles[:, 1:n_para(c)]') : c.particles[:, 1:n_para(c)]
get_loglh(c::Cloud) = c.particles[:, n_para(c) + 1]
get_logprior(c::Cloud) = c.particles[:, n_para(c) + 2]
get_old_loglh(c::Cloud) = c.particles[:, n_para(c) + 3]
get_logpost(c::Cloud) = get_loglh(c) .+ get_logprior(c)
get_accept(c::Cloud) = c.particles[:, n_para(c) + 4]
get_weights(c::Cloud) = c.particles[:, n_para(c) + 5]
function weighted_mean(c::Cloud)
    w = get_weights(c); X = c.particles[:, 1:n_para(c)]
    vec(sum(X .* w, dims = 1) ./ sum(w))
end
function weighted_cov(c::Cloud)                      # StatsBase.cov(X, Weights(w), corrected = false)
    w = get_weights(c) ./ sum(get_weights(c)); X = c.particles[:, 1:n_para(c)]
    Xc = X .- sum(X .* w, dims = 1)
    (Xc .* w)' * Xc
end
weighted_std(c::Cloud) = sqrt.(diag_of(weighted_cov(c)))
diag_of(A) = [A[i, i] for i in 1:size(A, 1)]

# ---- device likelihood families standing in for the closure (SMCMI_LIK_* in smcmi.h)
struct GaussIso; sigma::Float64; end
struct LinReg; sigma2::Float64; end
struct LinModel3; X::Matrix{Float64}; end
struct CapmLiteral; market::Matrix{Float64}; end
struct LGSSKalman; C::Matrix{Float64}; R::Matrix{Float64}; Z::Matrix{Float64}; kappa::Float64; end
const DeviceLikelihood = Union{GaussIso, LinReg, LinModel3, CapmLiteral, LGSSKalman}
family(::GaussIso) = Int32(0); family(::LinReg) = Int32(1); family(::LinModel3) = Int32(2); family(::CapmLiteral) = Int32(3)
family(::LGSSKalman) = Int32(4)
lik_par(l::GaussIso) = [l.sigma]; lik_par(l::LinReg) = [l.sigma2]; lik_par(l::LGSSKalman) = [l.kappa]; lik_par(::Any) = Float64[]
lik_aux(l::LinModel3) = l.X; lik_aux(l::CapmLiteral) = l.market
lik_aux(l::LGSSKalman) = reshape(vcat(vec(l.C'), vec(l.R'), vec(l.Z')), 1, :)          # [C 8x8 | R 8x3 | Z 3x8] row-major, 112 doubles
lik_aux(::Any) = zeros(0, 0)

prior_code(d::Normal) = (Int32(0), d.μ, d.σ)
prior_code(d::Uniform) = (Int32(1), d.a, d.b)
prior_code(d::Gamma) = (Int32(2), shape(d), scale(d))
prior_code(d::Beta) = (Int32(3), d.α, d.β)
prior_code(d::InverseGamma) = (Int32(4), shape(d), scale(d))
prior_code(d::ModelConstructors.RootInverseGamma) = (Int32(5), d.ν, d.τ)

const RESAMPLER = Dict(:systematic => Int32(0), :multinomial => Int32(1), :polyalgo => Int32(1))
const CALLBACK_ERRORS = (ModelConstructors.ParamBoundsError, LinearAlgebra.LAPACKException, LinearAlgebra.PosDefException,
                         LinearAlgebra.SingularException, DomainError)          # the five the reference maps to -Inf (mutation.jl:112-121)

# ---- the closure as a batch callback: int (*)(const double *theta, int64_t m, int64_t d, double *out, void *user_data)
mutable struct CallbackEnv      # (mutable: pointer_from_objref needs an object with a stable address)
    f::Function
    parameters::ParameterVector
    data::Matrix{Float64}
    toggle::Bool              # toggle_regime!(parameters, 1) after the evaluation (src/mutation.jl:98-110)
end
function lik_trampoline(theta::Ptr{Float64}, m::Int64, d::Int64, out::Ptr{Float64}, ud::Ptr{Cvoid})::Cint
    env = unsafe_pointer_to_objref(ud)::CallbackEnv
    try
        th = unsafe_wrap(Array, theta, (m, d))              # column-major m x d, as the C side packs it
        o = unsafe_wrap(Array, out, (m,))
        p = deepcopy(env.parameters)
        for k in 1:m
            o[k] = try
                update!(p, th[k, :])                        # mutation.jl:93 (cannot throw: only in-bounds proposals arrive)
                v = env.f(p, env.data)                      # mutation.jl:96
                env.toggle && ModelConstructors.toggle_regime!(p, 1)      # mutation.jl:98-100, 108-110
                v
            catch err
                isa(err, Union{CALLBACK_ERRORS...}) ? -Inf : rethrow(err)
            end
        end
        return Cint(0)
    catch err
        @error "SMCMI: the likelihood callback threw" exception = (err, catch_backtrace())
        return Cint(1)                                      # -> SMCMI_ERR_CALLBACK, nothing unwinds through the C frames
    end
end

function set_model!(h::Handle, parameters, lik, data, which::Integer, keep::Vector{Any}; toggle::Bool = false)
    if lik === nothing
        check(ccall((:smcmi_set_likelihood, LIB), Cint,
                    (Handle, Int32, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64),
                    h, which, -1, C_NULL, 0, C_NULL, 0, 0, C_NULL, 0, 0))
    elseif lik isa DeviceLikelihood
        par = lik_par(lik); aux = lik_aux(lik)
        check(ccall((:smcmi_set_likelihood, LIB), Cint,
                    (Handle, Int32, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int64),
                    h, which, family(lik), par, length(par), data, size(data, 1), size(data, 2), aux, size(aux, 1), size(aux, 2)))
    else
        env = CallbackEnv(lik, parameters, data, toggle)
        push!(keep, env)                                    # rooted for the lifetime of the handle
        fptr = @cfunction(lik_trampoline, Cint, (Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Cvoid}))
        check(ccall((:smcmi_set_likelihood_callback, LIB), Cint, (Handle, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                    h, which, fptr, pointer_from_objref(env)))
    end
end

function create(n_parts, d, seed, device, max_stages)
    h = Ref{Handle}(C_NULL)
    check(ccall((:smcmi_create, LIB), Cint, (Ref{Config}, Ref{Handle}), Config(n_parts, n_parts, 0, d, device, seed, max_stages, 1), h))
    h[]
end
destroy(h::Handle) = ccall((:smcmi_destroy, LIB), Cint, (Handle,), h)

# One entry per column of cloud.particles: (fixed, value, bounds, prior).  With regime_switching = true the regime-1 values of all
# parameters come first, then, parameter by parameter, the values of regimes 2, 3, ... (src/smc_main.jl:207-234: n_para counts them,
# para_symbols names them key_reg<i>) - the layout ModelConstructors.update!(parameters, θ) expects for a regime-switching vector.
function flat_entries(parameters, regime_switching::Bool)
    ent = Any[(p.fixed, p.value, p.valuebounds, p.fixed ? nothing : p.prior.value) for p in parameters]
    if regime_switching
        for p in parameters
            isempty(p.regimes) && continue
            for i in 2:length(p.regimes[:value])
                fx = haskey(p.regimes, :fixed) && haskey(p.regimes[:fixed], i) ? p.regimes[:fixed][i] : p.fixed
                vb = haskey(p.regimes, :valuebounds) && haskey(p.regimes[:valuebounds], i) ? p.regimes[:valuebounds][i] : p.valuebounds
                pr = haskey(p.regimes, :prior) && haskey(p.regimes[:prior], i) ? p.regimes[:prior][i].value : (p.fixed ? nothing : p.prior.value)
                push!(ent, (fx, p.regimes[:value][i], vb, fx ? nothing : pr))
            end
        end
    end
    ent
end

function set_parameters!(h::Handle, parameters; regime_switching::Bool = false)
    ent = flat_entries(parameters, regime_switching)
    fixed = Int32[e[1] ? 1 : 0 for e in ent]
    lo = Float64[e[3][1] for e in ent]; hi = Float64[e[3][2] for e in ent]
    codes = [e[1] ? (Int32(0), e[2], 1.0) : prior_code(e[4]) for e in ent]
    fam = Int32[x[1] for x in codes]; pa = Float64[x[2] for x in codes]; pb = Float64[x[3] for x in codes]
    check(ccall((:smcmi_set_parameters, LIB), Cint, (Handle, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
                h, fixed, lo, hi, fam, pa, pb))
end

# initial_draw! (src/initialization.jl:88-119): the device samples every prior family (Normal, Uniform, Gamma, Beta, InverseGamma,
# RootInverseGamma) with the bounds redraw of rand(parameters), scores the draws (device family, or the closure through the callback)
# and redraws particles without a finite log-likelihood (:23-63)
initial_draw!(h::Handle, parameters, lik, n_parts::Int, d::Int) = check(ccall((:smcmi_init_from_prior, LIB), Cint, (Handle,), h))

stage_path(savepath, i) = replace(savepath, ".jld2" => "_stage=$(i).jld2")         # smc_main.jl:500

"""
    smc(loglikelihood, parameters, data; kwargs...)

Drop-in for `SMC.smc` (src/smc_main.jl:118-161): same positional arguments, same keyword list, same files - `savepath` receives
`cloud`, `w`, `W` (JLD2) and `particle_store_path` the `smcparams` matrix (HDF5), intermediate saves go to
`savepath` with `_stage=i` (smc_main.jl:499-507, 513-526).  `loglikelihood` is the user's closure
`loglikelihood(parameters::ParameterVector, data::Matrix{Float64})::Float64` or one of the `DeviceLikelihood` structs.
Returns `(cloud, w, W)` in addition to writing the files (the reference returns nothing, quirk Q8).
`regime_switching = true` samples the extra regime values as additional columns (a device family then takes the flattened vector); `parallel` is moot (the device is
the parallelism).
"""
function smc(loglikelihood, parameters::ParameterVector, data::Matrix{Float64};
             verbose::Symbol = :low, testing::Bool = false, data_vintage::String = Dates.format(today(), "yymmdd"),
             parallel::Bool = false, n_parts::Int = 5_000, n_blocks::Int = 1, n_mh_steps::Int = 1,
             λ::Float64 = 2.1, n_Φ::Int64 = 300, resampling_method::Symbol = :systematic, threshold_ratio::Float64 = 0.5,
             c::Float64 = 0.5, α::Float64 = 1.0, target::Float64 = 0.25,
             use_fixed_schedule::Bool = true, tempering_target::Float64 = 0.97,
             old_data::Matrix{Float64} = Matrix{Float64}(undef, size(data, 1), 0), old_cloud = Cloud(0, 0),     # (any Cloud-shaped struct: SMC.Cloud too)
             old_loglikelihood = loglikelihood, old_vintage::String = "", smc_iteration::Int = 1,
             run_test::Bool = false, filestring_addl::Vector{String} = Vector{String}(),
             loadpath::String = "", savepath::String = "smc_cloud.jld2", particle_store_path::String = "smcsave.h5",
             save_intermediate::Bool = false, intermediate_stage_increment::Int = 10, continue_intermediate::Bool = false,
             intermediate_stage_start::Int = 0, tempered_update_prior_weight::Float64 = 0.0,
             regime_switching::Bool = false, toggle::Bool = true, debug_assertion::Bool = false,
             log_prob_old_data::Float64 = 0.0, seed::Integer = rand(UInt64), device::Integer = 0)
    haskey(RESAMPLER, resampling_method) || throw("Invalid resampler in SMC. Options are :systematic, :multinomial, or :polyalgo")
    0.0 <= tempered_update_prior_weight <= 1.0 ||
        throw(DomainError("The keyword tempered_update_prior_weight must be within the interval [0, 1] but " *
                          "is currently set to $(tempered_update_prior_weight)"))
    d = length(flat_entries(parameters, regime_switching))                                         # n_para incl. the regime columns, :207-216
    all(e -> e[1], flat_entries(parameters, regime_switching)) && throw(AssertionError("All model parameters are fixed!"))      # :236-239
    tempered_update = !isempty(old_data)
    haskey(VERBOSITY, verbose) || throw(ArgumentError("verbose must be one of :none, :low, :high"))
    old_cloud = as_cloud(old_cloud)
    max_stages = use_fixed_schedule ? n_Φ : 4 * n_Φ + 64          # (records + two N x max_stages history matrices; as the Python mirror)
    method = RESAMPLER[resampling_method]
    keep = Any[]                                             # callback environments (GC roots)
    h = create(n_parts, d, seed, device, max_stages)
    try
        set_parameters!(h, parameters; regime_switching = regime_switching)
        set_model!(h, parameters, loglikelihood, data, 0, keep; toggle = regime_switching && toggle)
        set_model!(h, parameters, tempered_update ? old_loglikelihood : nothing, old_data, 1, keep; toggle = regime_switching && toggle)
        initial_ess = 0.0
        W1 = nothing                                         # W_matrix[:, 1] of a tempered update (smc_main.jl:364-365)
        cont = false
        if tempered_update
            cloud0 = cloud_isempty(old_cloud) ? JLD2.load(loadpath, "cloud") : old_cloud              # :245-246
            initial_ess = tempered_cloud!(h, cloud0, parameters, old_loglikelihood, old_data, n_parts, tempered_update_prior_weight,
                                          method, seed, device, keep)
            P0 = Matrix{Float64}(undef, n_parts, d + 5)
            check(ccall((:smcmi_download_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, P0))
            W1 = P0[:, d + 5]
        elseif continue_intermediate
            cont = load_intermediate!(h, loadpath, n_Φ, λ, d)                                            # :334-335, 355-361
        else
            initial_draw!(h, parameters, loglikelihood, n_parts, d)
        end
        res = Result()
        while true
            stop = 0
            if save_intermediate
                ls = LoopState()
                cont && check(ccall((:smcmi_get_loop_state, LIB), Cint, (Handle, Ref{LoopState}), h, ls))
                i_now = cont ? Int(ls.stage_index) : 1
                stop = (div(i_now, intermediate_stage_increment) + 1) * intermediate_stage_increment
            end
            run_test && (stop = stop == 0 ? 3 : min(stop, 3))      # `if run_test && (i == 3) break` (smc_main.jl:495)
            rc = RunConfig(n_blocks, n_mh_steps, λ, n_Φ, method, threshold_ratio, c, α, target, use_fixed_schedule ? 1 : 0,
                           tempering_target, tempered_update_prior_weight, log_prob_old_data, 0, 0, 0, initial_ess, 0.0, stop, cont ? 1 : 0)
            check(ccall((:smcmi_run, LIB), Cint, (Handle, Ref{RunConfig}, Ref{Result}), h, rc, res))
            res.paused == 0 && break
            run_test && Int(res.n_stages) >= 3 && break
            cl, w, W, j = collect_cloud(h, n_parts, d, n_Φ, res)
            JLD2.jldopen(stage_path(savepath, cl.stage_index), true, true, true, JLD2.IOStream) do file          # :499-507
                write(file, "cloud", cl); write(file, "w", w); write(file, "W", W); write(file, "j", j)
            end
            cont = true
        end
        cloud, w, W, _ = collect_cloud(h, n_parts, d, n_Φ, res)
        W1 === nothing || (W[:, 1] = sum(W1) <= 1.0 ? W1 .* n_parts : W1)
        print_stages(h, cloud, parameters, regime_switching; verbose = verbose, use_fixed_schedule = use_fixed_schedule)
        if !testing                                                                                          # :513-526
            HDF5.h5open(particle_store_path, "w") do simfile
                simfile["smcparams"] = cloud.particles[:, 1:d]
            end
            JLD2.jldopen(savepath, true, true, true, JLD2.IOStream) do file
                write(file, "cloud", cloud); write(file, "w", w); write(file, "W", W)
            end
        end
        return cloud, w, W
    finally
        destroy(h)
    end
end

# verbose output (src/util.jl:117-180).  The loop runs on the device without a host round trip per stage, so the stage lines are printed
# after the run from the per-stage records (the fields end_stage_print shows: iteration, ϕ, c, acceptance rate, ESS, resamples so far);
# :high adds the final weighted means and standard deviations.  Per-stage wall times do not exist: the total is printed once.
const VERBOSITY = Dict(:none => 0, :low => 1, :high => 2)
# the getters copy as many records / history columns as the handle holds (smcmi_stages_held), whatever the caller expects
stages_held(h::Handle) = (k = Ref{Int32}(0); check(ccall((:smcmi_stages_held, LIB), Cint, (Handle, Ref{Int32}), h, k)); Int(k[]))
function stage_records(h::Handle)
    ns = stages_held(h)
    phi = Vector{Float64}(undef, ns); ess = similar(phi); cs = similar(phi); acc = similar(phi); rs = Vector{Int32}(undef, ns)
    check(ccall((:smcmi_get_stage_records, LIB), Cint, (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                h, phi, ess, cs, acc, rs))
    return phi, ess, cs, acc, rs
end
function print_stages(h::Handle, cloud::Cloud, parameters, regime_switching::Bool; verbose::Symbol = :low, use_fixed_schedule::Bool = true)
    VERBOSITY[verbose] >= VERBOSITY[:low] || return
    phi, ess, cs, acc, rs = stage_records(h)
    ns = min(cloud.stage_index, length(phi)); bar = "--------------------------"
    for i in 2:ns
        println(bar, "\n", use_fixed_schedule ? "Iteration = $(i) / $(cloud.n_Φ)" : "Iteration = $(i)", "\n", bar, "\nphi = $(phi[i])\n", bar)
        println("c = $(cs[i])\naccept = $(i < ns ? acc[i] : cloud.accept)\nESS = $(ess[i])   ($(sum(rs[2:i])) total resamples.)\n", bar)
    end
    println("time elapsed: $(round(cloud.total_sampling_time / 60, digits = 4)) minutes")
    if VERBOSITY[verbose] >= VERBOSITY[:high]
        μ = weighted_mean(cloud); σ = weighted_std(cloud)
        println("Mean and standard deviation of parameter estimates")
        for (n, sym) in enumerate(para_symbols(parameters, regime_switching))
            println("$(sym) = $(round(μ[n], digits = 5)), $(round(σ[n], digits = 5))")
        end
    end
end
# para_symbols (smc_main.jl:207-234): the keys, then key_reg<i> for the regime columns
function para_symbols(parameters, regime_switching::Bool)
    syms = Symbol[p.key for p in parameters]
    if regime_switching
        for p in parameters
            isempty(p.regimes) && continue
            for i in 2:length(p.regimes[:value])
                push!(syms, Symbol(p.key, "_reg", i))
            end
        end
    end
    syms
end

function collect_cloud(h::Handle, n_parts, d, n_Φ, res::Result)
    ns = Int(res.n_stages)
    cloud = Cloud(d, n_parts)
    check(ccall((:smcmi_download_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, cloud.particles))
    phi, ess = stage_records(h)
    cloud.tempering_schedule = phi[1:ns]; cloud.ESS = ess[1:ns]; cloud.stage_index = ns; cloud.n_Φ = n_Φ
    cloud.resamples = res.resamples; cloud.c = res.c; cloud.accept = res.accept; cloud.total_sampling_time = res.seconds
    w = Matrix{Float64}(undef, n_parts, stages_held(h)); W = similar(w)
    check(ccall((:smcmi_get_history, LIB), Cint, (Handle, Ptr{Float64}, Ptr{Float64}), h, w, W))
    w = w[:, 1:ns]; W = W[:, 1:ns]
    ls = LoopState()
    check(ccall((:smcmi_get_loop_state, LIB), Cint, (Handle, Ref{LoopState}), h, ls))
    return cloud, w, W, Int(ls.j)
end

# continue_intermediate (smc_main.jl:334-335, 355-361): cloud, w, W, j from the intermediate file back into the handle
function load_intermediate!(h::Handle, loadpath, n_Φ, λ, d)
    cloud = JLD2.load(loadpath, "cloud"); w = JLD2.load(loadpath, "w"); W = JLD2.load(loadpath, "W"); j = JLD2.load(loadpath, "j")
    ns = cloud.stage_index
    check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, cloud.particles))
    zs = zeros(ns)
    check(ccall((:smcmi_set_stage_records, LIB), Cint, (Handle, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                h, ns, cloud.tempering_schedule, cloud.ESS, zs, zs, zeros(Int32, ns)))
    check(ccall((:smcmi_set_history, LIB), Cint, (Handle, Int32, Ptr{Float64}, Ptr{Float64}), h, ns, w[:, 1:ns], W[:, 1:ns]))
    sched = ((collect(1:n_Φ) .- 1) ./ (n_Φ - 1)) .^ λ
    logmdd = sum(log.(vec(sum(w[:, 2:ns] .* W[:, 1:ns-1], dims = 1)) ./ length(cloud)))        # SURVEY §8 a-9
    ls = LoopState()
    ls.stage_index = ns; ls.j = j; ls.resampled_last_period = 0; ls.resamples = cloud.resamples
    ls.phi_n = cloud.tempering_schedule[ns]; ls.phi_prop = sched[j]; ls.c = cloud.c; ls.accept = cloud.accept
    ls.ess = cloud.ESS[ns]; ls.logmdd = logmdd
    check(ccall((:smcmi_set_loop_state, LIB), Cint, (Handle, Ref{LoopState}), h, ls))
    return true
end

# Initial cloud of a tempered update on the device (smc_main.jl:244-333).  Returns cloud.ESS[1] of the new run.
function tempered_cloud!(h::Handle, old::Cloud, parameters, old_lik, old_data, n_parts, pw, method, seed, device, keep)
    old_n = length(old); d = n_para(old)
    if pw == 0.0 && old_n == n_parts                                                             # :249-260
        check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, old.particles))
        check(ccall((:smcmi_initialize_likelihoods, LIB), Cint, (Handle,), h))
        return old.ESS[end]
    end
    n_to = Int(round((1 - pw) * n_parts)); n_pr = n_parts - n_to                               # :262-264
    if n_to > 0
        ho = create(old_n, d, seed, device, 2)
        try
            check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), ho, old.particles))
            check(ccall((:smcmi_bridge_resample, LIB), Cint, (Handle, Handle, Int32, UInt32, Int64, Ptr{Float64}, Ptr{Int64}),
                        h, ho, method, 0, n_to, C_NULL, C_NULL))                                 # :266-279
        finally
            destroy(ho)
        end
    end
    if n_pr > 0                                                                                  # :288-296
        hp = create(n_pr, d, seed, device, 2)
        try
            set_parameters!(hp, parameters)
            set_model!(hp, parameters, old_lik, old_data, 0, keep)
            set_model!(hp, parameters, nothing, old_data, 1, keep)
            initial_draw!(hp, parameters, old_lik, n_pr, d)
            check(ccall((:smcmi_copy_rows, LIB), Cint, (Handle, Int64, Handle, Int64, Int64), h, n_to, hp, 0, n_pr))
        finally
            destroy(hp)
        end
    end
    check(ccall((:smcmi_initialize_likelihoods, LIB), Cint, (Handle,), h))                       # :308
    check(ccall((:smcmi_normalize_weights, LIB), Cint, (Handle, Int32), h, 1))                   # :313-314
    check(ccall((:smcmi_resample, LIB), Cint, (Handle, Int32, UInt32, Ptr{Float64}, Ptr{Int64}), h, method, 1, C_NULL, C_NULL))   # :317-322
    return Float64(n_parts)                                                                      # :325
end

# ---- the other functions src/SMC.jl:13-17 exports, over the same C entry points (one-off handles; inside smc() these steps are fused
# kernels of the stage loop).  Random numbers come from the engine's Philox contract (DESIGN.md §2), keyed by `seed` / `stage`, not
# from Julia's global RNG.

"""
    resample(weights; n_parts = length(weights), method = :systematic, parallel = false, seed, stage)

src/resample.jl:23-72: the newly assigned (1-based) indices.  `n_parts != length(weights)` (the bridge case, smc_main.jl:266-279) goes
through `smcmi_bridge_resample`.
"""
function resample(weights::Vector{Float64}; n_parts::Int64 = length(weights), method::Symbol = :systematic, parallel::Bool = false,
                  seed::Integer = rand(UInt64), stage::Integer = 0, device::Integer = 0)
    haskey(RESAMPLER, method) || throw("Invalid resampler in SMC. Options are :systematic, :multinomial, or :polyalgo")
    n = length(weights)
    P = zeros(n, 1 + 5); P[:, 6] = weights
    h = create(n, 1, seed, device, 2)
    anc = Vector{Int64}(undef, n_parts)
    try
        check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, P))
        if n_parts == n
            check(ccall((:smcmi_resample, LIB), Cint, (Handle, Int32, UInt32, Ptr{Float64}, Ptr{Int64}), h, RESAMPLER[method], stage, C_NULL, anc))
        else
            hd = create(n_parts, 1, seed, device, 2)
            try
                check(ccall((:smcmi_bridge_resample, LIB), Cint, (Handle, Handle, Int32, UInt32, Int64, Ptr{Float64}, Ptr{Int64}),
                            hd, h, RESAMPLER[method], stage, n_parts, C_NULL, anc))
            finally
                destroy(hd)
            end
        end
    finally
        destroy(h)
    end
    return anc .+ 1
end

"""
    mvnormal_mixture_draw(θ_old, d_prop; c = 1.0, α = 1.0, seed, pid, stage)

src/helpers.jl:87-100: one draw from the three-component mixture around `θ_old` (`d_prop::MvNormal` = the proposal distribution
MvNormal(θ̄, Σ)).  Computed by `smcmi_propose` on a one-particle cloud, so the draw is the one the mutation kernel makes for particle
`pid` at `stage`.
"""
function mvnormal_mixture_draw(θ_old::Vector{Float64}, d_prop; c::Float64 = 1.0, α::Float64 = 1.0, seed::Integer = rand(UInt64),
                               pid::Integer = 0, stage::Integer = 0, device::Integer = 0)
    @assert 0 <= α <= 1
    d = length(θ_old); n = pid + 1
    μ = Vector{Float64}(d_prop.μ); Σ = Matrix{Float64}(d_prop.Σ)
    h = create(n, d, seed, device, 2)                       # (a cloud of pid + 1 rows: row pid + 1 carries the particle's RNG stream)
    try
        fixed = zeros(Int32, d); lo = fill(-1e300, d); hi = fill(1e300, d); fam = zeros(Int32, d); pa = zeros(d); pb = ones(d)
        check(ccall((:smcmi_set_parameters, LIB), Cint, (Handle, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
                    h, fixed, lo, hi, fam, pa, pb))
        set_model!(h, nothing, GaussIso(1.0), zeros(d, 1), 0, Any[])         # (a model must be set; the draw does not use it)
        P = zeros(n, d + 5); P[n, 1:d] = θ_old; P[:, d + 5] .= 1.0
        check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, P))
        prop = Matrix{Float64}(undef, n, d); lp = Vector{Float64}(undef, n); qd = Vector{Float64}(undef, n)
        bp = Int32[0, d]; bf = Int32.(collect(0:d-1))
        check(ccall((:smcmi_propose, LIB), Cint,
                    (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Int32, Int32, Int32, Float64, Float64, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    h, μ, collect(Σ'), bp, bf, 1, 0, 0, c, α, stage, prop, lp, qd))
        return prop[n, :]
    finally
        destroy(h)
    end
end

"""
    initial_draw!(loglikelihood, parameters, data, c::Cloud; parallel = false, regime_switching = false, toggle = true, seed)

src/initialization.jl:88-119: fills `c` with prior draws whose log-likelihood is finite (loglh, logprior; old_loglh = 0, weights = 1).
"""
function initial_draw!(loglikelihood, parameters::ParameterVector, data::Matrix{Float64}, c::Cloud; parallel::Bool = false,
                       regime_switching::Bool = false, toggle::Bool = true, seed::Integer = rand(UInt64), device::Integer = 0)
    n_parts = length(c); d = length(flat_entries(parameters, regime_switching))
    keep = Any[]
    h = create(n_parts, d, seed, device, 2)
    try
        set_parameters!(h, parameters; regime_switching = regime_switching)
        set_model!(h, parameters, loglikelihood, data, 0, keep; toggle = regime_switching && toggle)
        set_model!(h, parameters, nothing, data, 1, keep)
        initial_draw!(h, parameters, loglikelihood, n_parts, d)
        check(ccall((:smcmi_download_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, c.particles))
    finally
        destroy(h)
    end
    return nothing
end

"""
    mutation(loglikelihood, parameters, data, p, d_μ, d_Σ, n_free_para, blocks_free, blocks_all, ϕ_n, ϕ_n1; c, α, n_mh_steps,
             old_data, old_loglikelihood, regime_switching, toggle, seed, pid, stage)

src/mutation.jl:56-138 for ONE particle `p` (a row of `cloud.particles`): the updated particle.  Runs `smcmi_mutate` on a one-particle
shard with global id `pid`, i.e. exactly what the stage loop does to that particle at `stage` (blocks are 1-based index vectors, as
the reference passes them).
"""
function mutation(loglikelihood, parameters::ParameterVector, data::Matrix{Float64}, p::Vector{Float64}, d_μ::Vector{Float64},
                  d_Σ::Matrix{Float64}, n_free_para::Int, blocks_free::Vector{Vector{Int}}, blocks_all::Vector{Vector{Int}},
                  ϕ_n::Float64, ϕ_n1::Float64; c::Float64 = 1., α::Float64 = 1., n_mh_steps::Int = 1,
                  old_data::Matrix{Float64} = Matrix{Float64}(undef, size(data, 1), 0), old_loglikelihood = loglikelihood,
                  regime_switching::Bool = false, toggle::Bool = true, seed::Integer = rand(UInt64), pid::Integer = 0, stage::Integer = 0,
                  device::Integer = 0)
    d = length(p) - 5; n = pid + 1
    keep = Any[]
    h = create(n, d, seed, device, 2)                       # (row pid + 1 of a pid + 1 row cloud: the particle's own RNG stream)
    try
        set_parameters!(h, parameters; regime_switching = regime_switching)
        tempered = !isempty(old_data)
        P = zeros(n, d + 5); P[n, :] = p; P[1:n-1, d + 2] .= -Inf
        bp = Int32[0]; bf = Int32[]
        for b in blocks_free
            append!(bf, Int32.(b .- 1)); push!(bp, Int32(length(bf)))
        end
        nb = length(blocks_free)
        if loglikelihood isa DeviceLikelihood
            set_model!(h, parameters, loglikelihood, data, 0, keep)
            set_model!(h, parameters, tempered ? old_loglikelihood : nothing, old_data, 1, keep)
            check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, P))
            acc = Ref{Float64}(0.0)
            check(ccall((:smcmi_mutate, LIB), Cint,
                        (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Int32, Float64, Float64, Float64, Float64, Int32, UInt32, Ref{Float64}),
                        h, d_μ, collect(d_Σ'), bp, bf, nb, ϕ_n, ϕ_n1, c, α, n_mh_steps, stage, acc))
        else
            # a closure: the propose -> (host evaluates) -> accept split of include/smcmi.h, step by step and block by block
            set_model!(h, parameters, GaussIso(1.0), zeros(d, 1), 0, keep)                  # (placeholder family: the split never evaluates it)
            set_model!(h, parameters, nothing, old_data, 1, keep)
            check(ccall((:smcmi_upload_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, P))
            prop = Matrix{Float64}(undef, n, d); lp = Vector{Float64}(undef, n); qd = Vector{Float64}(undef, n)
            pv = deepcopy(parameters)
            function score(f, θ, dat)
                try
                    update!(pv, θ)
                    v = f(pv, dat)
                    regime_switching && toggle && ModelConstructors.toggle_regime!(pv, 1)
                    v
                catch err
                    isa(err, Union{CALLBACK_ERRORS...}) ? -Inf : rethrow(err)
                end
            end
            for step in 0:n_mh_steps-1, b in 0:nb-1
                check(ccall((:smcmi_propose, LIB), Cint,
                            (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Int32, Int32, Int32, Float64, Float64, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                            h, d_μ, collect(d_Σ'), bp, bf, nb, b, step, c, α, stage, prop, lp, qd))
                ll = fill(-Inf, n); llo = zeros(n)
                if isfinite(lp[n])                           # (the bounds check passed: mutation.jl:93)
                    ll[n] = score(loglikelihood, prop[n, :], data)
                    tempered && (llo[n] = score(old_loglikelihood, prop[n, :], old_data))
                end
                last = (step == n_mh_steps - 1 && b == nb - 1) ? 1 : 0
                check(ccall((:smcmi_accept, LIB), Cint, (Handle, Ptr{Float64}, Ptr{Float64}, Float64, Int32, Int32, Int32, UInt32, Int32),
                            h, ll, tempered ? llo : C_NULL, ϕ_n, b, step, nb, stage, last))
            end
        end
        check(ccall((:smcmi_download_cloud, LIB), Cint, (Handle, Ptr{Float64}), h, P))
        return P[n, :]
    finally
        destroy(h)
    end
end

end # module
