# PioranHIP.jl — the Julia side of the drop-in boundary (ccall only; no CUDA.jl / AMDGPU.jl).
#
# This is the binding a Pioran.jl maintainer adds to route the ScalableGP log-likelihood to
# libpioran_hip.so.  It overrides nothing for non-Float64 element types, so ForwardDiff Duals
# (Turing/NUTS, test/test_likelihood.jl:55) keep flowing through the original Julia `logl`.
#
# NOTE: the build image has no `julia`, so this file has never been parsed by a Julia front-end; the identical C ABI is
# exercised from Python (pioran.jl_amd/_lib.py), from plain C (tests/cabi_driver.c) and by the GPU parity tests, and
# tests/test_host.py::test_julia_shim_matches_header checks every ccall here against include/pioran_hip.h (symbol,
# argument count, integer / pointer kinds).  See INTEGRATION.md.
module PioranHIP

using LinearAlgebra
using Pioran
import Pioran: log_likelihood, SumOfCelerite, SemiSeparable, CARMA, celerite_coefs
import ChainRulesCore

const LIB = get(ENV, "PIORAN_HIP_LIB", "libpioran_hip")
const ABI_VERSION = 7

# PIORAN_BACKEND=julia keeps every call on Pioran's own Julia code (the escape hatch a deployment wants when no GPU is
# visible or for A/B comparisons); anything else (default "hip") routes Float64 calls to libpioran_hip.so.
# (USE_HIP, MIN_ROWS, MIN_STEPS: read from ENV in __init__, i.e. when the package is LOADED — module-level initialisers run at precompile time
#  and would bake the build machine's environment into the package cache.  LIB stays a const: ccall wants a constant library name; set
#  PIORAN_HIP_LIB before precompiling, or put the library on the loader's path.)
const USE_HIP = Ref(true)
# A drop-in must not be slower than what it replaces.  One scalar evaluation on the GPU walks a short series as a serial chain at ~0.15 us per
# step whatever the term count, Pioran.logl on one CPU core costs ~0.04 us per step and ROW (N = 8192: j = 2 terms 0.60 ms, j = 4 1.31 ms, j = 8
# 2.6 ms on the bench host; the reference's own figure: 0.85 / 1.6 / 3.7 ms — benchmark/benchmarks.jl:76-91, bench.py
# "reference_benchmark_grid_N8192").  So scalar calls with fewer than PIORAN_HIP_MIN_ROWS rows (R = 2 J; default 9, i.e. up to four terms)
# stay on Pioran's own code — unless the series is long: from PIORAN_HIP_MIN_STEPS steps on (default 3072) the library's time-parallel
# family (celerite_tp.hip: segments of the series on different CUs; round 6: boundary phase as a scan) is ahead of one core at every term count
# (N = 8192: j = 2 0.19 ms, j = 4 0.18 ms; N = 4096: 0.21 / 0.17 ms against 0.29 / 0.60 on one core).  Batched calls (logpdf_batch and friends) always use the GPU.  PIORAN_HIP_MIN_ROWS = 0 sends everything to the GPU.
const MIN_ROWS = Ref(9)
const MIN_STEPS = Ref(3072)
# rows the kernels execute: two per term, one for a term with b = d = 0 (Exp / DRW terms: src/Exp.jl:29-33, the DRW half of DRWCelerite src/psd.jl:270-273)
active_rows(b, d) = 2 * length(b) - count(i -> iszero(b[i]) && iszero(d[i]), eachindex(b))
use_hip_scalar(nrows::Integer, nsteps::Integer) = USE_HIP[] && (nrows >= MIN_ROWS[] || nsteps >= MIN_STEPS[])

function __init__()
    USE_HIP[] = lowercase(get(ENV, "PIORAN_BACKEND", "hip")) != "julia"
    MIN_ROWS[] = parse(Int, get(ENV, "PIORAN_HIP_MIN_ROWS", "9"))
    MIN_STEPS[] = parse(Int, get(ENV, "PIORAN_HIP_MIN_STEPS", "3072"))
    USE_HIP[] || return
    v = ccall((:pioran_abi_version, LIB), Cint, ())
    v == ABI_VERSION || error("libpioran_hip.so has ABI version $v, PioranHIP.jl was written for $ABI_VERSION")
end

struct PioranHIPError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:pioran_strerror, LIB), Cstring, (Cint,), rc))
    throw(PioranHIPError(rc, msg))
end

# ---- context: one per Julia process / MPI rank / Distributed worker (device = rank % ngpu) ---------
# Julia does not order finalizers: a Dataset may be finalized after its Context.  pioran_dataset_destroy dereferences the
# context, so the Context keeps the handles of its live data sets and destroys THEM first; a Dataset whose context is
# already gone only forgets its handle.
mutable struct Context
    h::Ptr{Cvoid}
    datasets::Set{Ptr{Cvoid}}
    function Context(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pioran_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        ctx = new(r[], Set{Ptr{Cvoid}}())
        finalizer(close!, ctx)
        return ctx
    end
end

function close!(ctx::Context)
    ctx.h == C_NULL && return
    for d in ctx.datasets
        ccall((:pioran_dataset_destroy, LIB), Cint, (Ptr{Cvoid},), d)
    end
    empty!(ctx.datasets)
    ccall((:pioran_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), ctx.h)
    ctx.h = C_NULL
    return
end

# diagnostic switches of a context (pioran_ctx_set_option: "no_split", "workspace_limit_mb", "dense_old_chain", ...) and the FP64 FMA rate
# the device sustains right now (TFLOP/s; the measured ceiling of any FP64 vector kernel on this box)
set_option!(ctx::Context, key::AbstractString, value::Union{Nothing, AbstractString} = "1") =
    check(ccall((:pioran_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Cstring), ctx.h, key, value === nothing ? C_NULL : value))
function fp64_probe(ctx::Context = default_context(); waves_per_simd::Integer = 2, ms::Real = 10.0)
    out = Ref{Cdouble}(0.0)
    check(ccall((:pioran_ctx_fp64_probe, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ref{Cdouble}), ctx.h, waves_per_simd, ms, out))
    return out[]
end

const DEFAULT_CTX = Ref{Union{Nothing, Context}}(nothing)
default_context() = (DEFAULT_CTX[] === nothing && (DEFAULT_CTX[] = Context(parse(Int, get(ENV, "PIORAN_HIP_DEVICE", "0")))); DEFAULT_CTX[])

# ---- scalar drop-in: logl(a, b, c, d, τ, y, σ2)  (src/celerite_solver.jl:312-334) -------------------
function logl_hip(a::Vector{Float64}, b::Vector{Float64}, c::Vector{Float64}, d::Vector{Float64},
                  τ::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}; ctx = default_context())
    out = Ref{Cdouble}(NaN)
    status = Ref{Int32}(0)
    GC.@preserve a b c d τ y σ2 begin
        check(ccall((:pioran_celerite_logl, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ref{Int32}),
                    ctx.h, length(τ), length(a), a, b, c, d, τ, y, σ2, out, status))
    end
    # status 2: the reference throws DomainError from log(D[1] < 0) (src/celerite_solver.jl:126)
    status[] == 2 && throw(DomainError(out[], "log-likelihood is not finite: covariance not positive definite"))
    return out[]
end

# Float64-only methods: everything else (Duals, BigFloat, views ...) falls through to Pioran's own code.
# The three methods of src/celerite_solver.jl:262-294, same solver switch and error text.
function log_likelihood(cov::SumOfCelerite, τ::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}; solver = :celerite)
    (solver == :celerite || solver == :celerite_matrix) ||
        error("solver $solver not recognised, use either :celerite or :celerite_matrix")
    if use_hip_scalar(active_rows(cov.b, cov.d), length(τ)) && eltype(cov.a) === Float64
        return logl_hip(collect(cov.a), collect(cov.b), collect(cov.c), collect(cov.d), τ, y, σ2)
    end
    return Pioran.logl(cov.a, cov.b, cov.c, cov.d, τ, y, σ2)
end

# CARMA (:272-282) and any other SemiSeparable kernel (:284-294): celerite_coefs, then real(logl(...))
function _log_likelihood_coefs(cov, τ::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}, solver)
    (solver == :celerite || solver == :celerite_matrix) ||
        error("solver $solver not recognised, use either :celerite or :celerite_matrix")
    a, b, c, d = celerite_coefs(cov)
    if use_hip_scalar(active_rows(b, d), length(τ)) && all(v -> eltype(v) <: Union{Float64, ComplexF64}, (a, b, c, d)) && all(v -> all(iszero, imag.(v)), (a, b, c, d))
        return logl_hip(collect(Float64, real.(a)), collect(Float64, real.(b)), collect(Float64, real.(c)), collect(Float64, real.(d)), τ, y, σ2)
    end
    return real(Pioran.logl(a, b, c, d, τ, y, σ2))
end
log_likelihood(cov::CARMA, τ::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}; solver = :celerite) =
    _log_likelihood_coefs(cov, τ, y, σ2, solver)
log_likelihood(cov::SemiSeparable, τ::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}; solver = :celerite) =
    _log_likelihood_coefs(cov, τ, y, σ2, solver)

# Reverse rule for the scalar likelihood: Zygote / ReverseDiff-based samplers (and Turing with an rrule-aware backend) get
# the GPU gradient instead of pushing Duals through Pioran.logl.  Cotangents for (a, b, c, d, y, σ2); τ is data.
function ChainRulesCore.rrule(::typeof(logl_hip), a::Vector{Float64}, b::Vector{Float64}, c::Vector{Float64}, d::Vector{Float64},
                              τ::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}; ctx = default_context())
    ds = Dataset(τ, y, σ2; ctx = ctx)
    g = try
        logpdf_grad_batch(ds, reshape(a, :, 1), reshape(b, :, 1), c, d; series = true)
    finally
        close!(ds)
    end
    g.status[1] == 2 && throw(DomainError(g.logl[1], "log-likelihood is not finite: covariance not positive definite"))
    function logl_pullback(Δ)
        NT = ChainRulesCore.NoTangent()
        return (NT, Δ .* vec(g.grad_a), Δ .* vec(g.grad_b), Δ .* vec(g.grad_c), Δ .* vec(g.grad_d), NT, Δ .* vec(g.grad_y), Δ .* vec(g.grad_σ²))
    end
    return g.logl[1], logl_pullback
end

# ---- data set handle + batched entry (the reference has no batch dimension) ---------------------------
mutable struct Dataset
    h::Ptr{Cvoid}
    N::Int
    ctx::Context
    function Dataset(t::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64}; ctx = default_context())
        r = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve t y σ2 check(ccall((:pioran_dataset_create, LIB), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}), ctx.h, length(t), t, y, σ2, r))
        ds = new(r[], length(t), ctx)
        push!(ctx.datasets, ds.h)
        finalizer(close!, ds)
        return ds
    end
end

function close!(ds::Dataset)
    ds.h == C_NULL && return
    if ds.ctx.h != C_NULL && ds.h in ds.ctx.datasets      # context still alive: it has not destroyed this handle yet
        delete!(ds.ctx.datasets, ds.h)
        ccall((:pioran_dataset_destroy, LIB), Cint, (Ptr{Cvoid},), ds.h)
    end
    ds.h = C_NULL
    return
end

"""
    logpdf_batch(ds, A, B, c, d; μ, ν, Y, S2)

`A`, `B`: `J × nbatch` matrices (one column per draw, Julia column-major = the ABI's `[B][J]`);
`c`, `d`: length-`J` vectors shared by all draws (as produced by `approx`, src/psd.jl:250,266-267) or
`J × nbatch` matrices.  Returns `(logl::Vector{Float64}, status::Vector{Int32})`.
This is what an ultranest `vectorized=true` callback calls once per batch of live points.
"""
function logpdf_batch(ds::Dataset, A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64};
                      μ::Union{Nothing, Vector{Float64}} = nothing, ν::Union{Nothing, Vector{Float64}} = nothing,
                      Y::Union{Nothing, Matrix{Float64}} = nothing, S2::Union{Nothing, Matrix{Float64}} = nothing)
    J, nb = size(A)
    out = Vector{Float64}(undef, nb)
    status = zeros(Int32, nb)
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    GC.@preserve A B c d μ ν Y S2 out status begin
        check(ccall((:pioran_celerite_logl_batch, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint,
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}),
                    ds.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), p(Y), p(S2), out, status))
    end
    return out, status
end

"""
    logpdf_batch_shift(ds, A, B, c, d, shift; μ, ν)

The shifted log-flux models (docs/src/ultranest.md:199-205): `ds` holds the raw flux `y` and `yerr.^2`; draw `b` is
evaluated on `log.(y .- shift[b])` with variances `ν[b] .* yerr.^2 ./ (y .- shift[b]).^2`, transformed on the GPU.
"""
function logpdf_batch_shift(ds::Dataset, A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64},
                            shift::Vector{Float64}; μ::Union{Nothing, Vector{Float64}} = nothing,
                            ν::Union{Nothing, Vector{Float64}} = nothing)
    J, nb = size(A)
    out = Vector{Float64}(undef, nb)
    status = zeros(Int32, nb)
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    GC.@preserve A B c d μ ν shift out status begin
        check(ccall((:pioran_celerite_logl_batch_shift, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint,
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}),
                    ds.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), shift, out, status))
    end
    return out, status
end

"""
    logpdf_batch_theta(ds, model, θ, norm, f_min, f_max, n_components; basis, μ, ν, shift, ...)

θ → log L in one call: `approx` (src/psd.jl:214-289) runs on the GPU in front of the scan, so only the sampled parameters
cross the boundary.  `model`: `:SingleBendingPowerLaw` (`θ` is `3 × nbatch`: α₁, f₁, α₂) or `:DoubleBendingPowerLaw`
(`5 × nbatch`); `norm`: the `norm` argument of `approx` per draw; `basis`: `"SHO"` or `"DRWCelerite"`.
This is the whole body of an ultranest `vectorized=true` likelihood for the models of docs/src/ultranest.md.
"""
function logpdf_batch_theta(ds::Dataset, model::Symbol, θ::Matrix{Float64}, norm::Vector{Float64}, f_min::Real, f_max::Real,
                            n_components::Integer; basis::String = "SHO", is_integrated_power::Bool = true,
                            S_low::Real = 20.0, S_high::Real = 20.0, μ::Union{Nothing, Vector{Float64}} = nothing,
                            ν::Union{Nothing, Vector{Float64}} = nothing, shift::Union{Nothing, Vector{Float64}} = nothing,
                            qpo::Union{Nothing, Array{Float64, 3}} = nothing)   # 3 × n_qpo × nbatch: (S₀, f₀, Q) of the QPO features
    m = model === :SingleBendingPowerLaw ? 0 : model === :DoubleBendingPowerLaw ? 1 : error("model $model not supported on the device")
    bs = basis == "SHO" ? 0 : basis == "DRWCelerite" ? 1 : error("basis $basis not supported on the device")
    size(θ, 1) == (m == 0 ? 3 : 5) || error("θ must be $(m == 0 ? 3 : 5) × nbatch")
    nb = size(θ, 2)
    nq = qpo === nothing ? 0 : size(qpo, 2)
    out = Vector{Float64}(undef, nb)
    status = zeros(Int32, nb)
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    GC.@preserve θ norm μ ν shift qpo out status begin
        check(ccall((:pioran_logpdf_batch_theta, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Cint, Int64, Cint, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}),
                    ds.h, nb, m, n_components, bs, is_integrated_power ? 1 : 0, f_min, f_max, S_low, S_high, θ, norm,
                    p(μ), p(ν), p(shift), nq, p(qpo), out, status, C_NULL, C_NULL))
    end
    return out, status
end

# ---- in-process farm: ONE Julia process driving several GPUs (no Distributed/MPI launcher needed) ----------------------
mutable struct Farm
    h::Ptr{Cvoid}
    function Farm(devices::Vector{<:Integer}, t::Vector{Float64}, y::Vector{Float64}, σ2::Vector{Float64})
        r = Ref{Ptr{Cvoid}}(C_NULL)
        dev = Vector{Cint}(devices)
        GC.@preserve dev t y σ2 check(ccall((:pioran_farm_create, LIB), Cint,
            (Cint, Ptr{Cint}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}), length(dev), dev, length(t), t, y, σ2, r))
        f = new(r[])
        finalizer(x -> ccall((:pioran_farm_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), f)
        return f
    end
end

"""
    logpdf_batch(farm, A, B, c, d; μ, ν, shift)

Same arguments as the single-GPU `logpdf_batch`; the draws are cut into contiguous shards, one per listed device, every
device writes its slice of the result directly (draws are independent: no collective).
"""
function logpdf_batch(farm::Farm, A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64};
                      μ::Union{Nothing, Vector{Float64}} = nothing, ν::Union{Nothing, Vector{Float64}} = nothing,
                      shift::Union{Nothing, Vector{Float64}} = nothing,
                      Y::Union{Nothing, Matrix{Float64}} = nothing, S2::Union{Nothing, Matrix{Float64}} = nothing)
    J, nb = size(A)
    out = Vector{Float64}(undef, nb)
    status = zeros(Int32, nb)
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    if Y !== nothing   # per-draw series, N x nb (a column per draw): CustomMean models, y .- mean_b.(t)
        (S2 !== nothing && shift === nothing) || error("Y and S2 come together, and not with shift")
        GC.@preserve A B c d μ ν Y S2 out status begin
            check(ccall((:pioran_farm_logl_batch_series, LIB), Cint,
                        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint,
                         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}),
                        farm.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), Y, S2, out, status))
        end
        return out, status
    end
    GC.@preserve A B c d μ ν shift out status begin
        check(ccall((:pioran_farm_logl_batch, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint,
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}),
                    farm.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), p(shift), out, status))
    end
    return out, status
end

# ---- posterior mean and simulation: pred / sim of src/celerite_solver.jl:363-483, 515-549 ------------------------------------
"""
    predict_batch(ds, A, B, c, d, τ; μ, ν)

Posterior mean at the times `τ` for every draw (column) of `A`, `B`: what `mean(posterior(f(t, σ²), y), τ)`
(src/scalable_GP.jl:64-72) computes one draw at a time.  `c`, `d`: length-`J` vectors shared by the draws, or `J × nbatch` matrices
(posterior draws of QPO / CARMA / free Celerite models: all draws still go through every kernel in one launch, each with its own
tables).  Ascending `τ` takes the fused evaluation.  Returns an `length(τ) × nbatch` matrix and the status vector.
"""
function predict_batch(ds::Dataset, A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64},
                       τ::Vector{Float64}; μ::Union{Nothing, Vector{Float64}} = nothing, ν::Union{Nothing, Vector{Float64}} = nothing)
    J, nb = size(A)
    out = Matrix{Float64}(undef, length(τ), nb)      # column-major = the ABI's [B][M]
    status = zeros(Int32, nb)
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    GC.@preserve A B c d μ ν τ out status begin
        check(ccall((:pioran_celerite_predict, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                     Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}),
                    ds.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), length(τ), τ, out, status))
    end
    return out, status
end

# Float64 drop-in for Pioran.pred (predict(cov, τ, t, y, σ²) reaches it for every covariance type, :348-361)
function pred_hip(a::Vector{Float64}, b::Vector{Float64}, c::Vector{Float64}, d::Vector{Float64}, τ::Vector{Float64},
                  t::Vector{Float64}, y::Vector{Float64}, σ²::Vector{Float64}; ctx = default_context())
    ds = Dataset(t, y, σ²; ctx = ctx)
    try
        out, _ = predict_batch(ds, reshape(a, :, 1), reshape(b, :, 1), c, d, τ)
        return vec(out)
    finally
        close!(ds)          # release the device copy now, not at the next GC
    end
end

"""
    simulate_batch(A, B, c, d, t, σ², q)

GP realisations `y = L D^(1/2) q` for every column of `A`, `B` (`c`, `d` shared vectors or `J × nbatch` matrices) from the standard
normals `q` (`length(t) × nbatch`);
`simulate(rng, cov, t, σ²)` (src/celerite_solver.jl:497-513) is `simulate_batch(..., randn(rng, N, 1))`.
"""
function simulate_batch(A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64}, t::Vector{Float64},
                        σ²::Vector{Float64}, q::Matrix{Float64}; ctx = default_context())
    J, nb = size(A)
    size(q) == (length(t), nb) || error("q must be length(t) × nbatch")
    out = Matrix{Float64}(undef, length(t), nb)
    GC.@preserve A B c d t σ² q out begin
        check(ccall((:pioran_celerite_simulate, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                    ctx.h, length(t), nb, J, A, B, c, d, c isa Vector ? 1 : 0, t, σ², q, out))
    end
    return out
end

# ---- value and gradient (reverse mode through the recurrence on the GPU) --------------------------------------------------
"""
    logpdf_grad_batch(ds, A, B, c, d; μ, ν, series = false, cd = true)

log L and ∂log L/∂(a_j, b_j, c_j, d_j) (`J × nbatch` each), ∂/∂ν, ∂/∂μ for every draw; with `series = true` also ∂/∂y_n and
∂/∂σ²_n (`N × nbatch`).  `cd = false` leaves out ∂/∂(c_j, d_j) — all an `approx`-based model needs, whose `(c, d)` are fixed by the
spectral grid (6.2 instead of 7.0 ms at N = 1e4, J = 20; `grad_c`, `grad_d` are then `nothing`).  `c`, `d`: length-`J` vectors shared by the draws or `J × nbatch` matrices (QPO features, CARMA,
free Celerite terms).  This is what a `ChainRulesCore.rrule` / `LogDensityProblems.logdensity_and_gradient` for the GP
likelihood returns instead of pushing ForwardDiff Duals through `Pioran.logl` (test/test_likelihood.jl:55-60); the chain
rule from (a, b, c, d) to the PSD parameters goes through `approx`, which stays in Julia.
"""
function logpdf_grad_batch(ds::Dataset, A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64};
                           μ::Union{Nothing, Vector{Float64}} = nothing, ν::Union{Nothing, Vector{Float64}} = nothing,
                           series::Bool = false, cd::Bool = true)
    J, nb = size(A)
    out = Vector{Float64}(undef, nb); status = zeros(Int32, nb)
    ga = Matrix{Float64}(undef, J, nb); gb = Matrix{Float64}(undef, J, nb)
    gc = cd ? Matrix{Float64}(undef, J, nb) : nothing; gd = cd ? Matrix{Float64}(undef, J, nb) : nothing
    gν = Vector{Float64}(undef, nb); gμ = Vector{Float64}(undef, nb)
    gy = series ? Matrix{Float64}(undef, ds.N, nb) : nothing
    gs = series ? Matrix{Float64}(undef, ds.N, nb) : nothing
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    GC.@preserve A B c d μ ν out status ga gb gc gd gν gμ gy gs begin
        check(ccall((:pioran_celerite_logl_grad, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Cdouble}),
                    ds.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), out, status, ga, gb, p(gc), p(gd), gν, gμ, p(gy), p(gs)))
    end
    return (logl = out, status = status, grad_a = ga, grad_b = gb, grad_c = gc, grad_d = gd, grad_ν = gν, grad_μ = gμ,
            grad_y = gy, grad_σ² = gs)
end

"""
    logpdf_grad_batch_shift(ds, A, B, c, d, shift; μ, ν)

Value and gradient for the shifted log-flux models (docs/src/turing.md:205-230): `ds` holds the raw flux and `yerr.^2`;
returns in addition `grad_shift` = ∂log L/∂c per draw.
"""
function logpdf_grad_batch_shift(ds::Dataset, A::Matrix{Float64}, B::Matrix{Float64}, c::VecOrMat{Float64}, d::VecOrMat{Float64},
                                 shift::Vector{Float64}; μ::Union{Nothing, Vector{Float64}} = nothing,
                                 ν::Union{Nothing, Vector{Float64}} = nothing)
    J, nb = size(A)
    out = Vector{Float64}(undef, nb); status = zeros(Int32, nb)
    ga = Matrix{Float64}(undef, J, nb); gb = Matrix{Float64}(undef, J, nb)
    gc = Matrix{Float64}(undef, J, nb); gd = Matrix{Float64}(undef, J, nb)
    gν = Vector{Float64}(undef, nb); gμ = Vector{Float64}(undef, nb); gs = Vector{Float64}(undef, nb)
    p(x) = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    GC.@preserve A B c d μ ν shift out status ga gb gc gd gν gμ gs begin
        check(ccall((:pioran_celerite_logl_grad_shift, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                     Ptr{Cdouble}, Ptr{Cdouble}),
                    ds.h, nb, J, A, B, c, d, c isa Vector ? 1 : 0, p(μ), p(ν), shift, out, status, ga, gb, gc, gd, gν, gμ, gs))
    end
    return (logl = out, status = status, grad_a = ga, grad_b = gb, grad_c = gc, grad_d = gd, grad_ν = gν, grad_μ = gμ, grad_shift = gs)
end

# ---- dense solver: log_likelihood_direct (src/direct_solver.jl:6-21), returns +NLL -------------------------
function log_likelihood_direct_hip(cov::SemiSeparable, t::Vector{Float64}, y::Vector{Float64}, σ²::Vector{Float64};
                                   ctx = default_context())
    a, b, c, d = map(v -> collect(Float64, real.(v)), celerite_coefs(cov))
    out = Ref{Cdouble}(NaN)
    info = Ref{Int32}(0)
    GC.@preserve a b c d t y σ² check(ccall((:pioran_dense_nll, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ref{Cdouble}, Ref{Int32}), ctx.h, length(t), length(a), a, b, c, d, t, y, σ², out, info))
    info[] != 0 && throw(LinearAlgebra.PosDefException(info[]))
    return out[]
end

# predict_cov (src/direct_solver.jl:28-69): posterior covariance at τ, the matrix behind cov / std / rand of a PosteriorGP
function predict_cov_hip(cov::SemiSeparable, τ::Vector{Float64}, t::Vector{Float64}, σ²::Vector{Float64}; ctx = default_context())
    a, b, c, d = map(v -> collect(Float64, real.(v)), celerite_coefs(cov))
    M = length(τ)
    out = Matrix{Float64}(undef, M, M)
    info = Ref{Int32}(0)
    GC.@preserve a b c d τ t σ² out check(ccall((:pioran_dense_predict_cov, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64,
         Ptr{Cdouble}, Ptr{Cdouble}, Ref{Int32}), ctx.h, length(t), length(a), a, b, c, d, t, σ², M, τ, out, info))
    info[] != 0 && throw(LinearAlgebra.PosDefException(info[]))
    return out
end

# predict_direct (src/direct_solver.jl:75-119): dense posterior mean (and covariance) — the reference's ground truth for `predict`
function predict_direct_hip(cov::SemiSeparable, τ::Vector{Float64}, t::Vector{Float64}, y::Vector{Float64}, σ²::Vector{Float64},
                            with_covariance::Bool = false; ctx = default_context())
    a, b, c, d = map(v -> collect(Float64, real.(v)), celerite_coefs(cov))
    M = length(τ)
    μ = Vector{Float64}(undef, M)
    K = with_covariance ? Matrix{Float64}(undef, M, M) : nothing
    info = Ref{Int32}(0)
    GC.@preserve a b c d τ t y σ² μ K check(ccall((:pioran_dense_predict, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Int32}), ctx.h, length(t), length(a), a, b, c, d, t, y, σ², M, τ, μ,
        K === nothing ? Ptr{Cdouble}(C_NULL) : pointer(K), info))
    info[] != 0 && throw(LinearAlgebra.PosDefException(info[]))
    return with_covariance ? (μ, K) : μ
end

end # module
