# QuantumPropagatorsHIPExt.jl -- package-extension glue that plugs libqprop_hip.so (C ABI in
# include/qprop.h) behind QuantumPropagators.jl's own interface:
#
#     propagate(Ψ, H, tlist; method = :ChebyHIP)      # or :NewtonHIP
#     p = init_prop(Ψ, H, tlist; method = :ChebyHIP);  prop_step!(p); reinit_prop!(p, Ψ); ...
#
# `H` may be anything the reference's own propagators take: a `Generator` (from `hamiltonian(...)` /
# `liouvillian(...)`), a tuple generator `(H0, (H1, ϵ1), ...)` -- including the `(H,)` of
# test/test_propagate.jl:22,157 --, a static `Operator`, or a plain matrix (src/controls.jl:442-475).
#
# `Ψ` may be a host vector -- then `p.state` is a host `Vector{ComplexF64}` (pinned; 16·N bytes come back
# over PCIe after every step, as the reference's interface implies) -- or a device-resident
# `HIPState(Ψ)`: then `p.state` IS the device vector, `prop_step!` returns it without any transfer, and
# the vector-space interface the reference requires of a state (src/interfaces/state.jl:92…:
# copyto!, lmul!, axpy!, dot, norm, fill!, similar, copy, zero) runs on the GPU.
#
# It follows the pattern of the reference's own third-party backend,
# ext/QuantumPropagatorsExponentialUtilitiesExt.jl:74-210 (a propagator struct, an `init_prop` method for
# a new `Val`, a `prop_step!` method).  STATUS: written against include/qprop.h and the reference sources
# cited inline; Julia is not installed in the build container, so this file has NOT been executed.  What IS
# checked mechanically: tests/test_julia_glue_signatures.py parses every `ccall` below and compares name,
# return type, arity and argument types with include/qprop.h.  The Python ctypes mirror
# (quantumpropagators.jl_amd/lib.py + propagator.py) calls exactly the same entry points and is what the
# test-suite runs on the GPU.
module QuantumPropagatorsHIPExt

using LinearAlgebra
using SparseArrays
using QuantumPropagators
using QuantumPropagators: PWCPropagator, _pwc_process_parameters, _pwc_advance_time!, _pwc_set_t!,
    _get_uniform_dt
using QuantumPropagators.Controls: get_controls, discretize, evaluate
using QuantumPropagators.Generators: Generator, Operator, hamiltonian
import QuantumPropagators: init_prop, prop_step!, reinit_prop!, set_state!, set_t!

const LIB = "libqprop_hip"     # quantumpropagators.jl_amd/lib/ on the loader path (or a JLL)

# qp_c128 of include/qprop.h: two doubles, passed by value -- the layout of ComplexF64
const C128 = ComplexF64

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

# One destructor per kind of handle.  `ccall` needs its `(name, library)` pair as a CONSTANT expression (a Symbol held in a
# struct field and resolved at run time is rejected: "first argument not a pointer or valid constant expression"), so the
# handle carries the destructor as a function value, not as a name.
_ctx_destroy(p::Ptr{Cvoid}) = ccall((:qp_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), p)
_state_destroy(p::Ptr{Cvoid}) = ccall((:qp_state_destroy, LIB), Cint, (Ptr{Cvoid},), p)
_matrix_destroy(p::Ptr{Cvoid}) = ccall((:qp_matrix_destroy, LIB), Cint, (Ptr{Cvoid},), p)
_operator_destroy(p::Ptr{Cvoid}) = ccall((:qp_operator_destroy, LIB), Cint, (Ptr{Cvoid},), p)
_cheby_destroy(p::Ptr{Cvoid}) = ccall((:qp_cheby_destroy, LIB), Cint, (Ptr{Cvoid},), p)
_newton_destroy(p::Ptr{Cvoid}) = ccall((:qp_newton_destroy, LIB), Cint, (Ptr{Cvoid},), p)

# opaque handle with a finalizer; `destroy` is one of the destructors above.  The library lets handles be destroyed in
# any order (a context's record outlives qp_ctx_destroy), so finalizers need no ordering; a destructor neither yields nor
# allocates on the Julia side, as a finalizer must not.
mutable struct Handle
    ptr::Ptr{Cvoid}
    destroy::Function
    keep::Any                      # objects that must outlive the handle (e.g. the context)
    function Handle(ptr::Ptr{Cvoid}, destroy::Function, keep = nothing)
        h = new(ptr, destroy, keep)
        finalizer(h) do x
            if x.ptr != C_NULL
                x.destroy(x.ptr)
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
    return Handle(out[], _ctx_destroy)
end

# one context per device and task tree, created on first use (contexts are cheap: a stream and two events)
const _CTX = Dict{Int,Handle}()
const _CTX_LOCK = ReentrantLock()        # propagators are created from many Julia threads (GRAPE / Krotov trajectories)
default_ctx(device::Integer = 0) = lock(_CTX_LOCK) do
    get!(() -> make_ctx(device), _CTX, Int(device))
end

# ------------------------------------------------------------------------------------------
# HIPState: a device-resident state vector   (SURVEY 8f N3; src/interfaces/state.jl:92…)
# ------------------------------------------------------------------------------------------
"""
    HIPState(Ψ::AbstractVector; device = 0)     # upload
    HIPState(undef, n; device = 0)

Complex state vector in HBM.  An `AbstractVector{ComplexF64}` whose vector-space operations -- the ones
`check_state` (src/interfaces/state.jl:92…) asks for and `cheby!` / `newton!` use (src/cheby.jl:146-148) --
are library calls on the device: `copyto!`, `lmul!`, `axpy!`, `dot`, `norm`, `fill!`, `similar`, `copy`,
`zero`, `+`, `-`, `*` by a number.  `Array(Ψ)` / `copyto!(::Vector, Ψ)` download; scalar `getindex` works (one
element, slow: for debugging and printing only).  Passing a `HIPState` to `init_prop` / `propagate` with
`method = :ChebyHIP | :NewtonHIP` keeps the propagator's `state` on the device: `prop_step!` returns it
without a transfer.
"""
mutable struct HIPState <: AbstractVector{ComplexF64}
    h::Handle
    ctx::Handle
    n::Int
end

function HIPState(::UndefInitializer, n::Integer; device::Integer = 0, ctx::Handle = default_ctx(device))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qp_state_create, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), ctx, n, out))
    return HIPState(Handle(out[], _state_destroy, ctx), ctx, Int(n))
end
function HIPState(Ψ::AbstractVector; device::Integer = 0, ctx::Handle = default_ctx(device))
    s = HIPState(undef, length(Ψ); ctx)
    copyto!(s, Ψ)
    return s
end

Base.size(s::HIPState) = (s.n,)
Base.length(s::HIPState) = s.n
Base.IndexStyle(::Type{HIPState}) = IndexLinear()
Base.similar(s::HIPState) = HIPState(undef, s.n; ctx = s.ctx)
Base.similar(s::HIPState, ::Type{ComplexF64}) = similar(s)
Base.similar(s::HIPState, ::Type{ComplexF64}, dims::Dims{1}) = HIPState(undef, dims[1]; ctx = s.ctx)
Base.copy(s::HIPState) = copyto!(similar(s), s)
Base.zero(s::HIPState) = fill!(similar(s), 0)
QuantumPropagators.Interfaces.supports_inplace(::Type{HIPState}) = true     # src/interfaces/supports_inplace.jl

# host <-> device
function Base.copyto!(dst::HIPState, src::AbstractVector)
    length(src) == dst.n || throw(DimensionMismatch("HIPState of length $(dst.n), source of length $(length(src))"))
    host = convert(Vector{ComplexF64}, src)
    GC.@preserve host check(ccall((:qp_state_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{C128}), dst.h, host))
    return dst
end
function Base.copyto!(dst::Vector{ComplexF64}, src::HIPState)
    length(dst) == src.n || throw(DimensionMismatch("HIPState of length $(src.n), destination of length $(length(dst))"))
    GC.@preserve dst check(ccall((:qp_state_download, LIB), Cint, (Ptr{Cvoid}, Ptr{C128}), src.h, dst))
    return dst
end
Base.Array(s::HIPState) = copyto!(Vector{ComplexF64}(undef, s.n), s)
Base.collect(s::HIPState) = Array(s)
Base.getindex(s::HIPState, i::Int) = Array(s)[i]      # one download per call: debugging / show only

# device <-> device: the six primitives of src/cheby.jl:146-148
function Base.copyto!(dst::HIPState, src::HIPState)
    check(ccall((:qp_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), dst.h, src.h))
    return dst
end
function LinearAlgebra.lmul!(a::Number, x::HIPState)
    check(ccall((:qp_scal, LIB), Cint, (Ptr{Cvoid}, C128), x.h, ComplexF64(a)))
    return x
end
LinearAlgebra.rmul!(x::HIPState, a::Number) = lmul!(a, x)
function LinearAlgebra.axpy!(a::Number, x::HIPState, y::HIPState)
    check(ccall((:qp_axpy, LIB), Cint, (C128, Ptr{Cvoid}, Ptr{Cvoid}), ComplexF64(a), x.h, y.h))
    return y
end
function Base.fill!(x::HIPState, a::Number)
    check(ccall((:qp_fill, LIB), Cint, (Ptr{Cvoid}, C128), x.h, ComplexF64(a)))
    return x
end
function LinearAlgebra.dot(x::HIPState, y::HIPState)
    out = Ref{ComplexF64}(0)
    check(ccall((:qp_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{C128}), x.h, y.h, out))
    return out[]
end
function LinearAlgebra.norm(x::HIPState)
    out = Ref{Cdouble}(0.0)
    check(ccall((:qp_norm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), x.h, out))
    return out[]
end
Base.:+(x::HIPState, y::HIPState) = axpy!(1, y, copy(x))
Base.:-(x::HIPState, y::HIPState) = axpy!(-1, y, copy(x))
Base.:*(a::Number, x::HIPState) = lmul!(a, copy(x))
Base.:*(x::HIPState, a::Number) = a * x
Base.:(==)(x::HIPState, y::HIPState) = Array(x) == Array(y)
Base.isapprox(x::HIPState, y::HIPState; kwargs...) = isapprox(Array(x), Array(y); kwargs...)

# ------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------
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
    return Handle(out[], _matrix_destroy, ctx)
end

# Generators.Operator on the device: lazy sum Σ c_l H_l, the first `length(ops) - ncoeffs` ops are drift
# terms with c = 1 (src/generators.jl:111-125, :635); mul! at :634-645 becomes qp_mul.
function make_operator(ctx::Handle, ops::AbstractVector, ncoeffs::Integer)
    mats = [make_matrix(ctx, O) for O in ops]
    ptrs = Ptr{Cvoid}[m.ptr for m in mats]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve mats ptrs check(ccall((:qp_operator_create, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        ctx, ptrs, length(ptrs), ncoeffs, 0, out))
    return Handle(out[], _operator_destroy, ctx)     # the library copied the matrices: `mats` may go
end

# evaluate!(op::Operator, generator, tlist, n; vals_dict)   src/generators.jl:757-766
function set_coeffs!(op::Handle, coeffs::Vector{ComplexF64})
    isempty(coeffs) && return nothing
    GC.@preserve coeffs check(ccall((:qp_operator_set_coeffs, LIB), Cint,
        (Ptr{Cvoid}, Ptr{C128}, Cint), op, coeffs, length(coeffs)))
end

# What laying the operator out on the device cost and how it is laid out now (include/qprop.h): host milliseconds of the
# latest / of all builds, re-layouts forced after creation -- a complex coefficient takes a Hermitian-packed operator back
# to plain row blocks, once, with a slower mat-vec from then on --, and the current device format.
function operator_build_info(op::Handle)
    out = zeros(Cdouble, 4)
    GC.@preserve out check(ccall((:qp_operator_build_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), op, out))
    return (build_ms = out[1], build_ms_total = out[2], relayouts = Int(out[3]), format = Int(out[4]))
end

# Strip-walk plan of a Hermitian-packed lattice operator (the fused Chebyshev term then runs csrc/kernels_walk.hip)
function operator_walk_info(op::Handle)
    out = zeros(Int64, 8)
    GC.@preserve out check(ccall((:qp_operator_walk_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), op, out))
    return (valid = out[1] != 0, near = Int(out[2]), far = Int(out[3]), diagonal = out[4] != 0, rows_per_step = Int(out[5]),
            first_block = out[6], end_block = out[7], edge_blocks = out[8])
end

# Does qp_cheby_step form TWO Chebyshev terms per pass over the matrix values for this operator (csrc/kernels_walk2.hip: lattice
# operators beyond the Infinity Cache; same results bit for bit) and how is that walk cut?
function operator_walk2_info(op::Handle)
    out = zeros(Int64, 8)
    GC.@preserve out check(ccall((:qp_operator_walk2_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), op, out))
    return (valid = out[1] != 0, first_block = out[2], end_block = out[3], edge_blocks = out[4], useful_rows_per_chunk = Int(out[5]),
            chunks_per_strip_step = Int(out[6]), steps_per_wavefront = Int(out[7]), segments = Int(out[8]))
end

# Will qp_cheby_step_batched take the LDS-staged 4 x 4 tiles for a panel of `batch` states (a lattice operator's interior rows; the rest goes
# to the row kernel; same results bit for bit)?
function operator_spmm_tiles(op::Handle, batch::Integer)
    out = zeros(Int64, 6)
    GC.@preserve out check(ccall((:qp_operator_spmm_tiles, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), op, batch, out))
    return (taken = out[1] != 0, tiles = out[2], rest_rows = out[3], rows_per_step = out[4], far = Int(out[5]), near = Int(out[6]))
end

# A qubit-register generator applied from its Pauli strings instead of stored matrices (include/qprop.h: qp_pauli_operator_create;
# csrc/engine_pauli.hip): `terms[l]` = the strings of H_l of the lazy sum as (amplitude, xmask, zmask) -- xmask = the qubits with X
# or Y, zmask = those with Z or Y, qubit i = bit i of the basis index --, the last `ncoeffs` terms carry the controls' coefficients
# (set_coeffs!), as for make_operator.  What the reference would be handed as SparseMatrixCSC terms (src/generators.jl:634-645).
struct PauliString                  # qp_pauli_string
    xmask::UInt64; zmask::UInt64
    coef::C128
    op::Cint
end

function make_pauli_operator(ctx::Handle, nqubits::Integer, terms::AbstractVector, ncoeffs::Integer)
    flat = PauliString[]
    for (l, strings) in enumerate(terms), (amp, xm, zm) in strings
        push!(flat, PauliString(UInt64(xm), UInt64(zm), ComplexF64(amp), Cint(l - 1)))
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve flat check(ccall((:qp_pauli_operator_create, LIB), Cint,
        (Ptr{Cvoid}, Cint, Ptr{PauliString}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        ctx, nqubits, flat, length(flat), length(terms), ncoeffs, out))
    return Handle(out[], _operator_destroy, ctx)
end

# long distance (rows) of a walk plan with one further pair beyond its far reach (three-dimensional grids), 0 if none
function operator_walk_long(op::Handle)
    n = Ref{Int64}(0)
    check(ccall((:qp_operator_walk_long, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), op, n))
    return n[]
end

# Why the fused Chebyshev term of an operator does not take the strip walk (include/qprop.h: QP_WALK_*): (code, sentence);
# code 0 = it does.  init_prop(...; method = :ChebyHIP) warns once for a large Hermitian operator that fell off the fast path.
function operator_walk_reason(op::Handle)
    code = Ref{Cint}(0)
    text = zeros(UInt8, 256)
    GC.@preserve text check(ccall((:qp_operator_walk_reason, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{UInt8}, Csize_t),
        op, code, text, length(text)))
    return Int(code[]), unsafe_string(pointer(text))
end

# an operator that is no lattice but has a fast path of its own has nothing to be told (the Python mirror: lib.py, _warn_if_off_the_walk):
# a dense or matrix-free format, a column-blocked mirror, a value dictionary, or row blocks that are all block maps (qubit registers)
function operator_has_own_fast_path(op::Handle)
    fmt = Ref{Cint}(0)
    check(ccall((:qp_operator_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Cint}), op, C_NULL, C_NULL, C_NULL, fmt))
    (fmt[] == 4 || fmt[] == 5) && return true          # QP_FMT_MATFREE, QP_FMT_DENSE
    operator_colblock_info(op).valid && return true
    ve = zeros(Int64, 6)
    GC.@preserve ve check(ccall((:qp_operator_value_encoding_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), op, ve))
    ve[1] != 0 && return true
    enc = zeros(Int64, 8)
    GC.@preserve enc check(ccall((:qp_operator_encoding_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), op, enc))
    return enc[4] > 0 && enc[1] + enc[2] == 0          # upper sections: block maps, none with per-entry int32 / int16 columns
end

# a context knob (include/qprop.h: qp_ctx_tuning_get)
function ctx_tuning_get(ctx::Handle, key::AbstractString)
    v = Ref{Cint}(0)
    check(ccall((:qp_ctx_tuning_get, LIB), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cint}), ctx, key, v))
    return Int(v[])
end

# column-blocked mirror of an operator with irregular columns (include/qprop.h: qp_operator_colblock_info):
# (has one, column blocks, log2 of the columns per block, rows per tile, longest segment, tiles, share of single-line gathers)
function operator_colblock_info(op::Handle)
    out = zeros(Int64, 6)
    share = Ref{Cdouble}(0.0)
    GC.@preserve out check(ccall((:qp_operator_colblock_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}), op, out, share))
    return (valid = out[1] != 0, column_blocks = out[2], log2_block_columns = out[3], rows_per_tile = out[4],
        longest_segment = out[5], tiles = out[6], own_line_share = share[])
end

# explicit zeros that qp_operator_create added to complete a lattice operator's rows (open boundaries of a grid)
function operator_fill_info(op::Handle)
    n = Ref{Int64}(0)
    check(ccall((:qp_operator_fill_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), op, n))
    return n[]
end

# mul!(C, A::Operator, B, α, β) on device vectors   src/generators.jl:634-645
function hip_mul!(y::HIPState, op::Handle, x::HIPState, α::Number = true, β::Number = false)
    check(ccall((:qp_mul, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, C128, C128),
        op, x.h, y.h, ComplexF64(α), ComplexF64(β)))
    return y
end

# What the reference accepts as a generator, as (ops, amplitudes, static coefficients):
#   Generator                 -> its ops and amplitudes                      (src/generators.jl:45-62)
#   tuple (H0, (H1, ϵ1), ...) -> hamiltonian(terms...) of the same terms     (src/controls.jl:442-463 sums the
#                                same matrices with the same coefficients; `(H,)` is the drift alone)
#   Operator                  -> ops with constant coefficients              (src/generators.jl:119-135)
#   a matrix                  -> one drift term                              (src/controls.jl:309-313, :466-475)
struct LazySum
    ops::Vector{Any}
    amplitudes::Vector{Any}
    coeffs::Vector{ComplexF64}     # constant coefficients of a static Operator (applied once)
end
lazy_sum(G::Generator) = LazySum(Any[G.ops...], Any[G.amplitudes...], ComplexF64[])
lazy_sum(O::Operator) = LazySum(Any[O.ops...], Any[], ComplexF64[O.coeffs...])
lazy_sum(A::AbstractMatrix) = LazySum(Any[A], Any[], ComplexF64[])
lazy_sum(terms::Tuple) = lazy_sum(hamiltonian(terms...; check = false))
lazy_sum(G) = throw(ArgumentError("the HIP backend needs a Generator, a tuple generator, an Operator or a matrix, not a $(typeof(G))"))

function device_operator(ctx::Handle, G::LazySum)
    ncoeffs = isempty(G.amplitudes) ? length(G.coeffs) : length(G.amplitudes)
    op = make_operator(ctx, G.ops, ncoeffs)
    isempty(G.coeffs) || set_coeffs!(op, G.coeffs)
    return op
end

# the coefficients of interval n: evaluate(ampl, tlist, n; vals_dict)   src/pwc_utils.jl:86-92
interval_coeffs(G::LazySum, tlist, n, vals_dict) =
    ComplexF64[evaluate(a, tlist, n; vals_dict) for a in G.amplitudes]

# Page-lock a host-resident state vector (src/propagator.jl:119-126 keeps `state` on the host) so that the
# hand-back after every step runs at PCIe speed; undone by `unpin!` in the finalizer of the propagator that
# owns the vector.  The registration is portable across devices and threads (qprop.h), so the finalizer may
# run anywhere.  Only the in-place mode benefits: inplace = false hands out a fresh, unpinned vector per step.
pin!(Ψ::Vector{ComplexF64}) =
    (GC.@preserve Ψ check(ccall((:qp_host_register, LIB), Cint, (Ptr{Cvoid}, Csize_t), pointer(Ψ), sizeof(Ψ))); Ψ)
unpin!(Ψ::Vector{ComplexF64}) =
    GC.@preserve Ψ ccall((:qp_host_unregister, LIB), Cint, (Ptr{Cvoid},), pointer(Ψ))

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
        s = state isa HIPState ? state : HIPState(Vector{ComplexF64}(state); ctx)
        m_max = get(kwargs, :m_max, 60)
        m_min = get(kwargs, :m_min, 25)
        lo = Ref{Cdouble}(0.0); hi = Ref{Cdouble}(0.0)
        check(ccall((:qp_specrange_arnoldi, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cdouble, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
            op, s.h, m_min, m_max, get(kwargs, :prec, 1e-3), get(kwargs, :norm_min, 1e-15),
            get(kwargs, :enlarge, true) ? 1 : 0, lo, hi))
        return lo[], hi[]
    else
        throw(ArgumentError("specrange method $method is not available on the HIP backend"))
    end
end

# ------------------------------------------------------------------------------------------
# propagators
# ------------------------------------------------------------------------------------------
# `state` is what the caller sees: a HIPState (device-resident: it IS `dstate`) or a host Vector mirror.
const StateT = Union{HIPState,Vector{ComplexF64}}

mutable struct ChebyHIPPropagator{GT} <: PWCPropagator
    const generator::GT
    state::StateT
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
    lazy::LazySum
    ctx::Handle
    op::Handle
    dstate::HIPState
    wrk::Handle
    coeffs::Vector{Float64}
    Δ::Float64
    E_min::Float64
    dt::Float64
    limit::Float64
end

set_t!(p::ChebyHIPPropagator, t) = _pwc_set_t!(p, t)

function _envelope(ctx, op, G::LazySum, N, control_ranges, method; kwargs...)
    # cheby_get_spectral_envelope   src/cheby_propagator.jl:331-345: all controls at their
    # minimum / maximum simultaneously; a generator without controls has one operator
    if isempty(G.amplitudes)
        return hip_specrange(ctx, op, N, method; kwargs...)
    end
    controls = collect(keys(control_ranges))
    amps(vals) = ComplexF64[evaluate(a, [0.0, 1.0], 1; vals_dict = vals) for a in G.amplitudes]
    set_coeffs!(op, amps(IdDict(c => control_ranges[c][2] for c in controls)))
    E_min, E_max = hip_specrange(ctx, op, N, method; kwargs...)
    set_coeffs!(op, amps(IdDict(c => control_ranges[c][1] for c in controls)))
    _E_min, _E_max = hip_specrange(ctx, op, N, method; kwargs...)
    return min(E_min, _E_min), max(E_max, _E_max)
end

# the propagator's state pair for a caller state: (what the caller sees, the device vector)
function _state_pair(ctx::Handle, state, inplace::Bool)
    if state isa HIPState
        d = inplace ? copy(state) : state            # init_prop copies when in-place  (src/cheby_propagator.jl:158)
        return d, d
    end
    Ψ = Vector{ComplexF64}(inplace ? copy(state) : state)
    return Ψ, HIPState(Ψ; ctx)
end

# init_prop(state, generator, tlist, ::Val{:Cheby}; …)   src/cheby_propagator.jl:87-175
function init_prop(state, generator, tlist, ::Val{:ChebyHIP};
                   inplace = true, backward = false, verbose = false, parameters = nothing,
                   control_ranges = nothing, specrange_method = :auto, specrange_buffer = 0.01,
                   cheby_coeffs_limit = 1e-12, check_normalization = false,
                   uniform_dt_tolerance = 1e-12, device = 0, specrange_kwargs...)
    tlist = convert(Vector{Float64}, tlist)
    G = lazy_sum(generator)
    controls = get_controls(generator)
    controlvals = [discretize(control, tlist) for control in controls]
    parameters = _pwc_process_parameters(parameters, controls, tlist)
    if isnothing(control_ranges)
        control_ranges = IdDict(c => (minimum(controlvals[i]), maximum(controlvals[i]))
                                for (i, c) in enumerate(controls))
    end
    ctx = state isa HIPState ? state.ctx : default_ctx(device)
    op = device_operator(ctx, G)
    N = length(state)
    if N >= 64 * ctx_tuning_get(ctx, "walk_min_blocks")      # large enough for the strip walk: say so if it is not taken
        code, why = operator_walk_reason(op)
        # 1 not Hermitian (newton! territory), 4 too few blocks, 16 switched off: nothing to report; neither for an operator
        # with a fast path of its own
        (code in (0, 1, 4, 16)) || operator_has_own_fast_path(op) ||
            @warn "ChebyHIP: the operator does not take the strip walk (up to 1.8x per term): $why" maxlog = 1
    end
    E_min, E_max = _envelope(ctx, op, G, N, control_ranges, specrange_method; specrange_kwargs...)
    Δ = E_max - E_min
    @assert Δ > 0.0
    δ = specrange_buffer * Δ                       # :131-133
    E_min = E_min - δ / 2
    Δ = Δ + δ
    dt = _get_uniform_dt(tlist; tol = uniform_dt_tolerance, warn = true)
    isnothing(dt) && error("Chebychev propagation only works on a uniform time grid")
    coeffs = hip_cheby_coeffs(Δ, dt, cheby_coeffs_limit)
    Ψ, dstate = _state_pair(ctx, state, inplace)
    wrk_out = Ref{Ptr{Cvoid}}(C_NULL)              # ChebyWrk, src/cheby.jl:87-124
    check(ccall((:qp_cheby_create, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), ctx, N, wrk_out))
    wrk = Handle(wrk_out[], _cheby_destroy, ctx)
    n = 1
    t = tlist[1]
    if backward
        n = length(tlist) - 1
        t = float(tlist[n+1])
    end
    p = ChebyHIPPropagator{typeof(generator)}(
        generator, Ψ, t, n, tlist, parameters, controls, control_ranges, backward, inplace,
        specrange_method, specrange_buffer, Dict{Symbol,Any}(specrange_kwargs), check_normalization,
        G, ctx, op, dstate, wrk, coeffs, Δ, E_min, dt, cheby_coeffs_limit)
    if inplace && Ψ isa Vector{ComplexF64}         # the in-place propagator owns its host mirror for its whole life: pin it
        pin!(Ψ)
        finalizer(q -> (s = getfield(q, :state); s isa Vector{ComplexF64} && unpin!(s)), p)
    end
    return p
end

# the state after a step, as the caller sees it: nothing to do for a device-resident state; the host
# mirror is refreshed (in place: the same, pinned, vector; not in place: a new vector per step)
function _hand_back!(p)
    p.state isa HIPState && return p.state
    if p.inplace
        copyto!(p.state, p.dstate)
    else
        setfield!(p, :state, Array(p.dstate))
    end
    return p.state
end

# prop_step!(::ChebyPropagator)   src/cheby_propagator.jl:348-386
function prop_step!(p::ChebyHIPPropagator)
    n = p.n
    tlist = getfield(p, :tlist)
    (0 < n < length(tlist)) || return nothing
    vals_dict = IdDict(c => p.parameters[c][n] for c in p.controls)        # _pwc_set_genop!  src/pwc_utils.jl:86-92
    set_coeffs!(p.op, interval_coeffs(p.lazy, tlist, n, vals_dict))
    dt = p.backward ? -p.dt : p.dt
    coeffs = p.coeffs
    if !p.inplace && p.state isa HIPState         # a new state object per step  (src/interfaces/propagator.jl:154-156)
        p.dstate = copy(p.dstate)
        setfield!(p, :state, p.dstate)
    end
    GC.@preserve coeffs check(ccall((:qp_cheby_step, LIB), Cint,                 # cheby!  src/cheby.jl:150-213
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint),
        p.wrk, p.op, p.dstate.h, coeffs, length(coeffs), p.Δ, p.E_min, dt, p.dt, p.limit,
        p.check_normalization ? 1 : 0))
    _hand_back!(p)
    _pwc_advance_time!(p)                                                        # src/pwc_utils.jl:102-112
    return p.state
end

# set_state!   src/propagator.jl:367-377
function set_state!(p::Union{ChebyHIPPropagator}, state)
    if p.state isa HIPState
        if state ≢ p.state
            if p.inplace
                copyto!(p.dstate, state)                # HIPState or host vector: device copy or upload
            else
                # not in place: the reference REBINDS propagator.state (src/propagator.jl:372-375); states the caller
                # already holds -- the object the last prop_step! returned -- must not be overwritten
                p.dstate = state isa HIPState ? copy(state) : HIPState(state; ctx = p.dstate.ctx)
                setfield!(p, :state, p.dstate)
            end
        end
        return p.state
    end
    if state ≢ p.state
        if p.inplace
            copyto!(p.state, state isa HIPState ? Array(state) : state)
        else
            setfield!(p, :state, state isa HIPState ? Array(state) : convert(Vector{ComplexF64}, state))
        end
    end
    copyto!(p.dstate, p.state)
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
        E_min, E_max = _envelope(p.ctx, p.op, p.lazy, length(p.state), ranges,
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
    ms_exposed::Cdouble
    sweeps_onepass::Cint; sweeps_onepass_redone::Cint
end

mutable struct NewtonHIPPropagator{GT} <: PWCPropagator
    const generator::GT
    state::StateT
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
    lazy::LazySum
    ctx::Handle
    op::Handle
    dstate::HIPState
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

function init_prop(state, generator, tlist, ::Val{:NewtonHIP};
                   inplace = true, backward = false, verbose = false, parameters = nothing,
                   m_max = 10, func = nothing, norm_min = 1e-14, relerr = 1e-12, max_restarts = 50,
                   device = 0, _...)
    inplace || error("The Newton propagator is only implemented in-place")     # src/newton_propagator.jl:94
    tlist = convert(Vector{Float64}, tlist)
    G = lazy_sum(generator)
    controls = get_controls(generator)
    parameters = _pwc_process_parameters(parameters, controls, tlist)
    ctx = state isa HIPState ? state.ctx : default_ctx(device)
    op = device_operator(ctx, G)
    Ψ, dstate = _state_pair(ctx, state, true)
    wrk_out = Ref{Ptr{Cvoid}}(C_NULL)                                            # NewtonWrk, src/newton.jl:23-60
    check(ccall((:qp_newton_create, LIB), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}), ctx, length(Ψ), m_max, wrk_out))
    wrk = Handle(wrk_out[], _newton_destroy, ctx)
    n = 1
    t = tlist[1]
    if backward
        n = length(tlist) - 1
        t = float(tlist[n+1])
    end
    p = NewtonHIPPropagator{typeof(generator)}(generator, Ψ, t, n, tlist, parameters, controls, backward, inplace,
        func, norm_min, relerr, max_restarts, G, ctx, op, dstate, wrk,
        Ref(NewtonStats(0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)))
    if Ψ isa Vector{ComplexF64}
        pin!(Ψ)
        finalizer(q -> (s = getfield(q, :state); s isa Vector{ComplexF64} && unpin!(s)), p)
    end
    return p
end

function prop_step!(p::NewtonHIPPropagator)
    n = p.n
    tlist = getfield(p, :tlist)
    (0 < n < length(tlist)) || return nothing
    dt = tlist[n+1] - tlist[n]                                                   # :127-130 (non-uniform grids allowed)
    p.backward && (dt = -dt)
    vals_dict = IdDict(c => p.parameters[c][n] for c in p.controls)
    set_coeffs!(p.op, interval_coeffs(p.lazy, tlist, n, vals_dict))
    if isnothing(p.func)
        func_id, cb, user, box = 0, C_NULL, C_NULL, nothing                     # QP_FUNC_EXPMI
    else
        box = Ref{Any}(p.func)
        cb = @cfunction(_func_trampoline, Cvoid, (Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{Cvoid}))
        func_id, user = 2, pointer_from_objref(box)                             # QP_FUNC_CALLBACK
    end
    GC.@preserve box check(ccall((:qp_newton_step, LIB), Cint,                   # newton!  src/newton.jl:246-385
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Ptr{NewtonStats}),
        p.wrk, p.op, p.dstate.h, dt, func_id, cb, user, p.norm_min, p.relerr, p.max_restarts, p.stats))
    _hand_back!(p)
    _pwc_advance_time!(p)
    return p.state
end

function set_state!(p::NewtonHIPPropagator, state)
    if p.state isa HIPState
        state ≢ p.state && copyto!(p.dstate, state)
        return p.state
    end
    state ≢ p.state && copyto!(p.state, state isa HIPState ? Array(state) : state)
    copyto!(p.dstate, p.state)
    return p.state
end

end # module
