# QuantumPropagatorsHIPExt.jl -- package-extension glue that plugs libqprop_hip.so (C ABI in
# include/qprop.h) behind QuantumPropagators.jl's own interface:
#
#     propagate(Ψ, H, tlist; method = :ChebyHIP)      # or :NewtonHIP
#     p = init_prop(Ψ, H, tlist; method = :ChebyHIP);  prop_step!(p); reinit_prop!(p, Ψ); ...
#
# It follows the pattern of the reference's own third-party backend,
# ext/QuantumPropagatorsExponentialUtilitiesExt.jl:74-210 (a propagator struct, an
# `init_prop` method for a new `Val`, a `prop_step!` method).  STATUS: written against
# include/qprop.h and the reference sources cited inline; Julia is not installed in the build
# container, so this file has NOT been executed.  The Python ctypes mirror
# (quantumpropagators.jl_amd/lib.py + propagator.py) calls exactly the same entry points and is
# what the test-suite runs.
module QuantumPropagatorsHIPExt

using LinearAlgebra
using SparseArrays
using QuantumPropagators
using QuantumPropagators: PWCPropagator, _pwc_process_parameters, _pwc_advance_time!, _pwc_set_t!,
    _get_uniform_dt
using QuantumPropagators.Controls: get_controls, discretize, evaluate
using QuantumPropagators.Generators: Generator, Operator
import QuantumPropagators: init_prop, prop_step!, reinit_prop!, set_state!, set_t!

const LIB = "libqprop_hip"     # quantumpropagators.jl_amd/lib/ on the loader path (or a JLL)

# ------------------------------------------------------------------------------------------
# status codes -> the reference's exception types (include/qprop.h, "status codes")
# ------------------------------------------------------------------------------------------
struct QPropHIPError <: Exception
    status::Cint
    msg::String
end

function check(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:qp_last_error, LIB), Cstring, ()))
    if status in (3, 4, 5, 6, 7)          # QP_E_DT_MISMATCH … QP_E_DIVDIFF_UNDERFLOW: the reference's @assert
        throw(AssertionError(msg))
    elseif status in (1, 11)              # QP_E_BAD_ARG, QP_E_M_MAX
        throw(ArgumentError(msg))
    else
        throw(QPropHIPError(status, msg))
    end
end

# opaque handle with a finalizer; `destroy` is the name of the C destructor
mutable struct Handle
    ptr::Ptr{Cvoid}
    destroy::Symbol
    keep::Any                      # objects that must outlive the handle (e.g. the context)
    function Handle(ptr::Ptr{Cvoid}, destroy::Symbol, keep = nothing)
        h = new(ptr, destroy, keep)
        finalizer(h) do x
            if x.ptr != C_NULL
                ccall((x.destroy, LIB), Cint, (Ptr{Cvoid},), x.ptr)
                x.ptr = C_NULL
            end
        end
        return h
    end
end
Base.unsafe_convert(::Type{Ptr{Cvoid}}, h::Handle) = h.ptr

function make_ctx(device::Integer = 0)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qp_ctx_create, LIB), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device, C_NULL, out))
    return Handle(out[], :qp_ctx_destroy)
end

# One term of the generator.  The library converts Julia's 1-based SparseMatrixCSC{ComplexF64,Int64}
# (src/generators.jl:473-486) to its device format; nothing is copied on the Julia side.
function make_matrix(ctx::Handle, A::AbstractMatrix)
    S = SparseMatrixCSC{ComplexF64,Int64}(sparse(A))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve S check(ccall((:qp_matrix_create, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        ctx, size(S, 1), size(S, 2), nnz(S), S.colptr, S.rowval, S.nzval,
        0,   # QP_VAL_C128
        1,   # QP_LAYOUT_CSC
        1,   # index_base
        0,   # QP_FMT_AUTO
        out))
    return Handle(out[], :qp_matrix_destroy, ctx)
end

# Generators.Operator on the device: lazy sum Σ c_l H_l, drift terms first
# (src/generators.jl:111-125); mul! at :634-645 becomes qp_mul.
function make_operator(ctx::Handle, ops::AbstractVector, ncoeffs::Integer)
    mats = [make_matrix(ctx, O) for O in ops]
    ptrs = Ptr{Cvoid}[m.ptr for m in mats]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve mats ptrs check(ccall((:qp_operator_create, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        ctx, ptrs, length(ptrs), ncoeffs, 0, out))
    return Handle(out[], :qp_operator_destroy, ctx)     # the library copied the matrices: `mats` may go
end

# evaluate!(op::Operator, generator, tlist, n; vals_dict)   src/generators.jl:757-766
function set_coeffs!(op::Handle, coeffs::Vector{ComplexF64})
    isempty(coeffs) && return nothing
    GC.@preserve coeffs check(ccall((:qp_operator_set_coeffs, LIB), Cint,
        (Ptr{Cvoid}, Ptr{ComplexF64}, Cint), op, coeffs, length(coeffs)))
end

# Page-lock the propagator's host-resident state vector (src/propagator.jl:119-126 keeps `state` on the
# host) so that the hand-back after every step runs at PCIe speed; undone by `unpin!` in the finalizer of
# the propagator that owns the vector.
pin!(Ψ::Vector{ComplexF64}) =
    (GC.@preserve Ψ check(ccall((:qp_host_register, LIB), Cint, (Ptr{Cvoid}, Csize_t), pointer(Ψ), sizeof(Ψ))); Ψ)
unpin!(Ψ::Vector{ComplexF64}) =
    GC.@preserve Ψ ccall((:qp_host_unregister, LIB), Cint, (Ptr{Cvoid},), pointer(Ψ))

function make_state(ctx::Handle, Ψ::Vector{ComplexF64})
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qp_state_create, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), ctx, length(Ψ), out))
    s = Handle(out[], :qp_state_destroy, ctx)
    upload!(s, Ψ)
    return s
end
upload!(s::Handle, Ψ::Vector{ComplexF64}) =
    GC.@preserve Ψ check(ccall((:qp_state_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), s, Ψ))
download!(Ψ::Vector{ComplexF64}, s::Handle) =
    GC.@preserve Ψ check(ccall((:qp_state_download, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), s, Ψ))

# cheby_coeffs(Δ, dt; limit)   src/cheby.jl:25-39
function hip_cheby_coeffs(Δ::Float64, dt::Float64, limit::Float64)
    n = Ref{Cint}(0)
    coeffs = Vector{Float64}(undef, 64)
    while true
        status = ccall((:qp_cheby_coeffs, LIB), Cint, (Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Ptr{Cint}),
                       Δ, dt, limit, coeffs, length(coeffs), n)
        if status == 0
            return resize!(coeffs, n[])
        elseif n[] > length(coeffs)
            resize!(coeffs, n[])
        else
            check(status)
        end
    end
end

# specrange(H, method; kwargs...)   src/specrad.jl:36-140, with :arnoldi on the device
function hip_specrange(ctx::Handle, op::Handle, N::Integer, method::Symbol; kwargs...)
    if method == :auto
        if haskey(kwargs, :E_min) && haskey(kwargs, :E_max)
            method = :manual
        else
            method = :arnoldi      # (:diag for N <= 32 is host-side dense eigvals in the reference; keep it there)
        end
    end
    if method == :manual
        return float(kwargs[:E_min]), float(kwargs[:E_max])
    elseif method == :arnoldi
        state = get(kwargs, :state, nothing)
        if isnothing(state)                              # random_state, src/specrad.jl:153-158
            state = rand(N) .* exp.((2π * im) .* rand(N))
            state ./= norm(state)
        end
        s = make_state(ctx, Vector{ComplexF64}(state))
        m_max = get(kwargs, :m_max, 60)
        m_min = get(kwargs, :m_min, 25)
        lo = Ref{Cdouble}(0.0); hi = Ref{Cdouble}(0.0)
        check(ccall((:qp_specrange_arnoldi, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cdouble, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
            op, s, m_min, m_max, get(kwargs, :prec, 1e-3), get(kwargs, :norm_min, 1e-15),
            get(kwargs, :enlarge, true) ? 1 : 0, lo, hi))
        return lo[], hi[]
    else
        throw(ArgumentError("specrange method $method is not available on the HIP backend"))
    end
end

# ------------------------------------------------------------------------------------------
# propagators
# ------------------------------------------------------------------------------------------
mutable struct ChebyHIPPropagator{GT} <: PWCPropagator
    const generator::GT
    state::Vector{ComplexF64}          # host mirror returned to callers (identical object when in-place)
    t::Float64
    n::Int64
    const tlist::Vector{Float64}
    parameters::AbstractDict
    controls
    control_ranges::AbstractDict
    backward::Bool
    inplace::Bool
    specrange_method::Symbol
    specrange_buffer::Float64
    specrange_options::Dict{Symbol,Any}
    check_normalization::Bool
    # device side
    ctx::Handle
    op::Handle
    dstate::Handle
    wrk::Handle
    coeffs::Vector{Float64}
    Δ::Float64
    E_min::Float64
    dt::Float64
    limit::Float64
    drift_offset::Int
end

set_t!(p::ChebyHIPPropagator, t) = _pwc_set_t!(p, t)

function _envelope(ctx, op, generator::Generator, N, control_ranges, method; kwargs...)
    # cheby_get_spectral_envelope   src/cheby_propagator.jl:331-345: all controls at their
    # minimum / maximum simultaneously
    controls = collect(keys(control_ranges))
    amps(vals) = ComplexF64[evaluate(a, [0.0, 1.0], 1; vals_dict = vals) for a in generator.amplitudes]
    set_coeffs!(op, amps(IdDict(c => control_ranges[c][2] for c in controls)))
    E_min, E_max = hip_specrange(ctx, op, N, method; kwargs...)
    set_coeffs!(op, amps(IdDict(c => control_ranges[c][1] for c in controls)))
    _E_min, _E_max = hip_specrange(ctx, op, N, method; kwargs...)
    return min(E_min, _E_min), max(E_max, _E_max)
end

# init_prop(state, generator, tlist, ::Val{:Cheby}; …)   src/cheby_propagator.jl:87-175
function init_prop(state, generator::Generator, tlist, ::Val{:ChebyHIP};
                   inplace = true, backward = false, verbose = false, parameters = nothing,
                   control_ranges = nothing, specrange_method = :auto, specrange_buffer = 0.01,
                   cheby_coeffs_limit = 1e-12, check_normalization = false,
                   uniform_dt_tolerance = 1e-12, device = 0, specrange_kwargs...)
    tlist = convert(Vector{Float64}, tlist)
    controls = get_controls(generator)
    controlvals = [discretize(control, tlist) for control in controls]
    parameters = _pwc_process_parameters(parameters, controls, tlist)
    if isnothing(control_ranges)
        control_ranges = IdDict(c => (minimum(controlvals[i]), maximum(controlvals[i]))
                                for (i, c) in enumerate(controls))
    end
    ctx = make_ctx(device)
    op = make_operator(ctx, generator.ops, length(generator.amplitudes))
    N = length(state)
    E_min, E_max = _envelope(ctx, op, generator, N, control_ranges, specrange_method; specrange_kwargs...)
    Δ = E_max - E_min
    @assert Δ > 0.0
    δ = specrange_buffer * Δ                       # :131-133
    E_min = E_min - δ / 2
    Δ = Δ + δ
    dt = _get_uniform_dt(tlist; tol = uniform_dt_tolerance, warn = true)
    isnothing(dt) && error("Chebychev propagation only works on a uniform time grid")
    coeffs = hip_cheby_coeffs(Δ, dt, cheby_coeffs_limit)
    Ψ = Vector{ComplexF64}(inplace ? copy(state) : state)
    dstate = make_state(ctx, Ψ)
    wrk_out = Ref{Ptr{Cvoid}}(C_NULL)              # ChebyWrk, src/cheby.jl:87-124
    check(ccall((:qp_cheby_create, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), ctx, N, wrk_out))
    wrk = Handle(wrk_out[], :qp_cheby_destroy, ctx)
    n = 1
    t = tlist[1]
    if backward
        n = length(tlist) - 1
        t = float(tlist[n+1])
    end
    p = ChebyHIPPropagator{typeof(generator)}(
        generator, Ψ, t, n, tlist, parameters, controls, control_ranges, backward, inplace,
        specrange_method, specrange_buffer, Dict{Symbol,Any}(specrange_kwargs), check_normalization,
        ctx, op, dstate, wrk, coeffs, Δ, E_min, dt, cheby_coeffs_limit,
        length(generator.ops) - length(generator.amplitudes))
    if inplace                                     # the in-place propagator owns Ψ for its whole life: pin it
        pin!(Ψ)
        finalizer(q -> unpin!(getfield(q, :state)), p)
    end
    return p
end

# prop_step!(::ChebyPropagator)   src/cheby_propagator.jl:348-386
function prop_step!(p::ChebyHIPPropagator)
    n = p.n
    tlist = getfield(p, :tlist)
    (0 < n < length(tlist)) || return nothing
    generator = getfield(p, :generator)
    vals_dict = IdDict(c => p.parameters[c][n] for c in p.controls)        # _pwc_set_genop!  src/pwc_utils.jl:86-92
    set_coeffs!(p.op, ComplexF64[evaluate(a, tlist, n; vals_dict) for a in generator.amplitudes])
    dt = p.backward ? -p.dt : p.dt
    coeffs = p.coeffs
    GC.@preserve coeffs check(ccall((:qp_cheby_step, LIB), Cint,                 # cheby!  src/cheby.jl:150-213
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint),
        p.wrk, p.op, p.dstate, coeffs, length(coeffs), p.Δ, p.E_min, dt, p.dt, p.limit,
        p.check_normalization ? 1 : 0))
    # hand the state back (16·N bytes per step; a device-resident state type is row N3 of SURVEY 8f)
    if p.inplace
        download!(p.state, p.dstate)
    else
        Ψ = similar(p.state)
        download!(Ψ, p.dstate)
        setfield!(p, :state, Ψ)
    end
    _pwc_advance_time!(p)                                                        # src/pwc_utils.jl:102-112
    return p.state
end

# set_state!   src/propagator.jl:367-377
function set_state!(p::Union{ChebyHIPPropagator}, state)
    if state ≢ p.state
        if p.inplace
            copyto!(p.state, state)
        else
            setfield!(p, :state, convert(Vector{ComplexF64}, state))
        end
    end
    upload!(p.dstate, p.state)
    return p.state
end

# reinit_prop!(::ChebyPropagator, state; …)   src/cheby_propagator.jl:243-299
function reinit_prop!(p::ChebyHIPPropagator, state; transform_control_ranges = (c, lo, hi, check) -> (lo, hi), _...)
    set_state!(p, state)
    ranges = IdDict(c => (minimum(p.parameters[c]), maximum(p.parameters[c])) for c in p.controls)
    need = any(p.controls) do c
        lo, hi = transform_control_ranges(c, ranges[c][1], ranges[c][2], true)
        (lo < p.control_ranges[c][1]) || (hi > p.control_ranges[c][2])
    end
    if need
        for c in p.controls
            ranges[c] = transform_control_ranges(c, ranges[c][1], ranges[c][2], false)
        end
        E_min, E_max = _envelope(p.ctx, p.op, getfield(p, :generator), length(p.state), ranges,
                                 p.specrange_method; p.specrange_options...)
        Δ = E_max - E_min
        @assert Δ > 0.0
        δ = p.specrange_buffer * Δ
        p.E_min = E_min - δ / 2
        p.Δ = Δ + δ
        p.dt = float(p.tlist[2] - p.tlist[1])
        p.control_ranges = ranges
        p.coeffs = hip_cheby_coeffs(p.Δ, p.dt, p.limit)
    end
    _pwc_set_t!(p, float(p.backward ? p.tlist[end] : p.tlist[1]))
end

# ------------------------------------------------------------------------------------------
# Newton   src/newton_propagator.jl:62-153, src/newton.jl:246-385
# ------------------------------------------------------------------------------------------
struct NewtonStats                  # qp_newton_stats
    restarts::Cint; n_a::Cint; n_leja::Cint; m_last::Cint; n_matvec::Cint
    radius::Cdouble; last_relerr::Cdouble; norm_psi::Cdouble
    ms_arnoldi::Cdouble; ms_eig::Cdouble; ms_leja::Cdouble; ms_coeffs::Cdouble; ms_poly::Cdouble; ms_update::Cdouble
end

mutable struct NewtonHIPPropagator{GT} <: PWCPropagator
    const generator::GT
    state::Vector{ComplexF64}
    t::Float64
    n::Int64
    const tlist::Vector{Float64}
    parameters::AbstractDict
    controls
    backward::Bool
    inplace::Bool
    func
    norm_min::Float64
    relerr::Float64
    max_restarts::Int64
    ctx::Handle
    op::Handle
    dstate::Handle
    wrk::Handle
    stats::Base.RefValue{NewtonStats}
end

set_t!(p::NewtonHIPPropagator, t) = _pwc_set_t!(p, t)

# user `func` (any Julia callable z -> f(z)) through the C callback: QP_FUNC_CALLBACK
function _func_trampoline(z::Ptr{ComplexF64}, out::Ptr{ComplexF64}, user::Ptr{Cvoid})::Cvoid
    f = unsafe_pointer_to_objref(user)::Base.RefValue{Any}
    unsafe_store!(out, ComplexF64(f[](unsafe_load(z))))
    return nothing
end

function init_prop(state, generator::Generator, tlist, ::Val{:NewtonHIP};
                   inplace = true, backward = false, verbose = false, parameters = nothing,
                   m_max = 10, func = nothing, norm_min = 1e-14, relerr = 1e-12, max_restarts = 50,
                   device = 0, _...)
    inplace || error("The Newton propagator is only implemented in-place")     # src/newton_propagator.jl:94
    tlist = convert(Vector{Float64}, tlist)
    controls = get_controls(generator)
    parameters = _pwc_process_parameters(parameters, controls, tlist)
    ctx = make_ctx(device)
    op = make_operator(ctx, generator.ops, length(generator.amplitudes))
    Ψ = Vector{ComplexF64}(copy(state))
    dstate = make_state(ctx, Ψ)
    wrk_out = Ref{Ptr{Cvoid}}(C_NULL)                                            # NewtonWrk, src/newton.jl:23-60
    check(ccall((:qp_newton_create, LIB), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}), ctx, length(Ψ), m_max, wrk_out))
    wrk = Handle(wrk_out[], :qp_newton_destroy, ctx)
    n = 1
    t = tlist[1]
    if backward
        n = length(tlist) - 1
        t = float(tlist[n+1])
    end
    return NewtonHIPPropagator{typeof(generator)}(generator, Ψ, t, n, tlist, parameters, controls, backward, inplace,
        func, norm_min, relerr, max_restarts, ctx, op, dstate, wrk,
        Ref(NewtonStats(0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)))
end

function prop_step!(p::NewtonHIPPropagator)
    n = p.n
    tlist = getfield(p, :tlist)
    (0 < n < length(tlist)) || return nothing
    dt = tlist[n+1] - tlist[n]                                                   # :127-130 (non-uniform grids allowed)
    p.backward && (dt = -dt)
    generator = getfield(p, :generator)
    vals_dict = IdDict(c => p.parameters[c][n] for c in p.controls)
    set_coeffs!(p.op, ComplexF64[evaluate(a, tlist, n; vals_dict) for a in generator.amplitudes])
    if isnothing(p.func)
        func_id, cb, user, box = 0, C_NULL, C_NULL, nothing                     # QP_FUNC_EXPMI
    else
        box = Ref{Any}(p.func)
        cb = @cfunction(_func_trampoline, Cvoid, (Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{Cvoid}))
        func_id, user = 2, pointer_from_objref(box)                             # QP_FUNC_CALLBACK
    end
    GC.@preserve box check(ccall((:qp_newton_step, LIB), Cint,                   # newton!  src/newton.jl:246-385
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Ptr{NewtonStats}),
        p.wrk, p.op, p.dstate, dt, func_id, cb, user, p.norm_min, p.relerr, p.max_restarts, p.stats))
    download!(p.state, p.dstate)
    _pwc_advance_time!(p)
    return p.state
end

function set_state!(p::NewtonHIPPropagator, state)
    state ≢ p.state && copyto!(p.state, state)
    upload!(p.dstate, p.state)
    return p.state
end

end # module
