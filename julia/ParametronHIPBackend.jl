# ParametronHIPBackend.jl — routes Parametron.jl's update! hot path (src/model.jl:132-143) through libparametron_hip.so.
#
#     using Parametron, OSQP
#     include("ParametronHIP.jl"); include("ParametronHIPBackend.jl"); using .ParametronHIPBackend
#     model = Model(OSQP.Optimizer()); ... @objective / @constraint as in README.md:23-57 ...
#     hm = HIPModel(model)            # analyses the optimised lazy-expression DAGs once, allocates the plan, records the tape
#     solve!(hm)                      # = solve!(model) with update!(model) replaced by the device path
#
# How it hooks in.  The reference keeps every objective / constraint as a WrappedExpression — a FunctionWrapper around the optimised
# LazyExpression (src/lazyexpression.jl:65-71); the wrapped object stays reachable as `expr.f.obj[]` (the reference's own `show` uses it,
# :44-46), and an optimised node is `LazyExpression(Functions.matvecmul!, dest, A, x)` etc. (:200-302): the builder function and its
# arguments are plain fields.  `analyse` walks that DAG and turns the shapes of README Example 1 and 2 into device nodes:
#     matvecmul!(dest, A, x::Vector{Variable})          A a Parameter{Matrix} or the adjoint node of one (:206-217)
#     vecsubtract!/vecadd!(dest, <that>, b::Parameter)  -> DenseAffine(A, x, b, -1|+1), kept implicit
#     vecdot!(dest, r, r) / matvecmul!(dest::QuadraticFunction, r', r)   -> least-squares objective of a DenseAffine
#     copyto!/convert wrappers                          -> looked through
#     vecsubtract!/vecadd!(dest, x::Vector{Variable}, l::Parameter)     -> bounds rows `x - l` (test/model.jl:162-163)
#     scale!(dest, s::Parameter{<:Number}, <DenseAffine>)               -> the affine block times a scalar Parameter (:284-290)
#     vcat!(dest, <DenseAffine | bounds>...)                            -> one MOI function, the pieces stacked (:276-278)
#     bilinearmul!(dest, Q, x', y)                                      -> transpose(x) * Q * y objectives (:219-226)
# Anything else is an ERROR by default: `HIPModel(model)` (strict = true) throws an ArgumentError that names the record and the builder the
# analysis does not know — a model either runs on the device path or it does not start.  `HIPModel(model; strict = false)` keeps the
# reference's own CPU `update!` for such a record instead (logged once); `on_device(hm)` says which record runs where.  Never a silent
# approximation, and with strict = true never a silent CPU solve either.
#
# Zero allocation in steady state, as the reference promises (`@allocated solve!(model) == 0`, README.md:8,138, test/model.jl:116-124):
# update! below touches preallocated buffers only — Parameter values are copied into PAGE-LOCKED staging arrays (pmt_host_alloc) and
# travel on the plan's copy stream (two slots, pmt_plan_stage_upload[_2d]) while the previous re-evaluation is still running; the scalar
# the objective's constant is fetched into lives in the record.  tests/test_cabi_exports.py scans `refresh!` / `update!` / `solve!` for
# allocating constructors.
#
# `HIPModel(model; handoff = :host_csc, solver_update = f)`: instead of 252 MB of MOI terms (config 2) the HOST solver is handed what its
# own update takes — P's and A's CSC values, q, l, u (84 MB) in page-locked arrays that are filled WHILE the contraction runs (recorded
# fetches + pmt_quad_gram_csc_deliver_f64): `f(Px, Ax, q, l, u)` is e.g. `(a...) -> OSQP.update!(osqp; Px = a[1], Ax = a[2], q = a[3],
# l = a[4], u = a[5])`.  5.9 -> 1.73 ms per solve! at config 2 through the Python host of this repository (DESIGN.md §8.1; the link's floor
# for 84 MB is 1.55 ms).
#
# NOT EXECUTED in this repository's CI (no julia in the build image; tests/test_gpu_julia.py runs julia/example1_parity.jl when a julia
# binary is present on the GPU box).  tests/test_cabi_exports.py checks every ccall of julia/*.jl against include/parametron_hip.h —
# symbol, argument count and argument type classes.
module ParametronHIPBackend

export HIPModel, solve!, on_device

import Parametron
import Parametron: Model, Parameter, Variable, LazyExpression, setdirty!
import Parametron.Functions
import MathOptInterface
const MOI = MathOptInterface
using LinearAlgebra
using SparseArrays

import ..ParametronHIP
const H = ParametronHIP
const DevPtr = H.DevPtr

# ---------------------------------------------------------------------------------------------------------------------------
# device mirrors of Parameters (src/parameter.jl:36-104): the callback stays a host function; its value is uploaded when it ran

mutable struct DeviceParameter
    param::Parameter
    buf::DevPtr
    rows::Int
    cols::Int            # 0 for vectors, -1 for scalars
    ld::Int              # leading dimension of the device copy (rows rounded up to 16, + 64 when a multiple of 512: DESIGN.md §2)
    host::Vector{Float64}             # page-locked staging copy of the value (dense, column-major): the asynchronous upload reads it
    staging::NTuple{2, DevPtr}        # one device staging buffer per slot (pmt_plan_stage_slot)
    nnz::Int             # > 0: a SparseMatrixCSC with a FIXED pattern — buf / host / staging hold nzval only (BASELINE config 5)
    # SMALL models (HIPModel: small): the value travels through a page-locked MAILBOX in the device layout that the tape's first entries copy
    # from inside the one launch; the library writes it from the Parameter's own array (pmt_model_update, H.ModelRun)
    mailbox::Vector{Float64}
    slot::Int                         # slot in the ModelRun (-1: not registered)
    host_ptr::Ptr{Float64}            # address of the value array the slot currently reads (an out-of-place callback returns a new one)
    scalar::Vector{Float64}           # a Number's value, as the one-element array the slot reads
end

padded_rows(r) = r >= 64 ? 16 * cld(r, 16) : r
padded_ld(r) = (p = padded_rows(r); p >= 512 && p % 512 == 0 ? p + 64 : p)

function DeviceParameter(plan::H.Plan, p::Parameter)
    val = p()
    if val isa SparseMatrixCSC{Float64, Int64}
        r, c = size(val)
        nz = length(val.nzval)
        bytes = 8 * max(nz, 1)
        return DeviceParameter(p, H.alloc(plan, bytes), r, c, r, H.host_alloc(max(nz, 1)), (H.alloc(plan, bytes), H.alloc(plan, bytes)), nz, Float64[], -1, Ptr{Float64}(C_NULL), Float64[])
    elseif val isa AbstractMatrix
        r, c = size(val)
        ld = padded_ld(r)
        bytes = 8 * ld * max(c, 1)
        return DeviceParameter(p, H.alloc(plan, bytes), r, c, ld, H.host_alloc(r * c), (H.alloc(plan, bytes), H.alloc(plan, bytes)), 0, Float64[], -1, Ptr{Float64}(C_NULL), Float64[])     # zero filled: the padding rows stay zero
    elseif val isa AbstractVector
        r = length(val)
        bytes = 8 * max(padded_rows(r), 1)
        return DeviceParameter(p, H.alloc(plan, bytes), r, 0, padded_rows(r), H.host_alloc(r), (H.alloc(plan, bytes), H.alloc(plan, bytes)), 0, Float64[], -1, Ptr{Float64}(C_NULL), Float64[])
    elseif val isa Number
        return DeviceParameter(p, H.alloc(plan, 8), 1, -1, 1, H.host_alloc(1), (H.alloc(plan, 8), H.alloc(plan, 8)), 0, Float64[], -1, Ptr{Float64}(C_NULL), Float64[])
    end
    throw(ArgumentError("Parameters of type $(typeof(val)) are not supported on the device"))
end

"""evaluate the Parameter (runs its update function if dirty, src/parameter.jl:93-99), copy the value into its page-locked staging array
(`copyto!`: no allocation, whatever array type the user's callback fills) and start the upload on the plan's COPY stream into staging slot
`slot`; `commit!` consumes it on the plan's stream"""
function refresh!(plan::H.Plan, d::DeviceParameter, slot::Int)
    val = d.param()
    if d.nnz > 0
        length(val.nzval) == d.nnz || throw(DimensionMismatch("the sparsity pattern of a sparse Parameter must stay fixed"))
        copyto!(d.host, val.nzval)                      # the pattern is fixed (checked by length; rowval / colptr are the plan's)
        H.stage_upload!(plan, d.staging[slot + 1], d.host)
    elseif d.cols > 0
        copyto!(d.host, val)
        H.stage_upload_pitched!(plan, d.staging[slot + 1], d.ld, d.host, d.rows, d.cols)
    elseif d.cols == 0
        copyto!(d.host, val)
        H.stage_upload!(plan, d.staging[slot + 1], d.host)
    else
        d.host[1] = val
        H.stage_upload!(plan, d.staging[slot + 1], d.host)
    end
    nothing
end

function commit!(plan::H.Plan, d::DeviceParameter, slot::Int)
    H.commit_staged!(plan, d.buf, d.staging[slot + 1], d.nnz > 0 ? 8 * d.nnz : (d.cols > 0 ? 8 * d.ld * d.cols : 8 * max(d.ld, 1)))
    nothing
end

# ---------------------------------------------------------------------------------------------------------------------------
# analysis of the optimised DAG

unwrap(e) = e
unwrap(e::LazyExpression{<:Parametron.FunctionWrapper}) = unwrap(e.f.obj[])        # WrappedExpression -> the wrapped LazyExpression
function unwrap(e::LazyExpression)
    # look through identity / convert / copyto! wrappers (src/lazyexpression.jl:65-71, 280-282; src/moi_interop.jl:121,151)
    if e.f === identity && length(e.args) == 1
        return unwrap(e.args[1])
    elseif e.f === convert && length(e.args) == 2
        return unwrap(e.args[2])
    elseif e.f === copyto! && length(e.args) == 2
        return unwrap(e.args[2])
    end
    e
end

struct Unsupported <: Exception
    what::String
end

"A*x (+|-) b kept implicit: nothing but A and b is ever read (the variable of a term is its column's)"
struct DenseAffine
    A::Parameter
    transposed::Bool          # the node is A' * x (README Example 2: X' * g)
    x::Vector{Variable}
    b::Union{Nothing, Parameter}
    sign::Int                 # -1: A*x - b, +1: A*x + b, 0: no b
end

function dense_matrix(arg)
    arg isa Parameter && return arg, false
    e = unwrap(arg)
    if e isa LazyExpression && length(e.args) == 2 && e.args[2] isa Parameter && e.args[1] isa AbstractMatrix && !(e.f isa Function && e.f === Functions.matvecmul!)
        # the adjoint rule: LazyExpression(closure, dest::Matrix, A::Parameter) (src/lazyexpression.jl:206-217)
        return e.args[2], true
    end
    throw(Unsupported("matrix operand $(typeof(arg))"))
end

function analyse_affine(arg)::DenseAffine
    e = unwrap(arg)
    e isa LazyExpression || throw(Unsupported("affine operand $(typeof(e))"))
    if e.f === Functions.matvecmul! && length(e.args) == 3 && e.args[3] isa Vector{Variable}
        A, t = dense_matrix(e.args[2])
        return DenseAffine(A, t, e.args[3], nothing, 0)
    elseif (e.f === Functions.vecsubtract! || e.f === Functions.vecadd!) && length(e.args) == 3 && e.args[3] isa Parameter
        inner = analyse_affine(e.args[2])
        inner.b === nothing || throw(Unsupported("nested vecadd!/vecsubtract!"))
        return DenseAffine(inner.A, inner.transposed, inner.x, e.args[3], e.f === Functions.vecsubtract! ? -1 : 1)
    end
    throw(Unsupported("builder $(e.f)"))
end

"C*x (+|-) d for a SparseMatrixCSC Parameter C with a fixed pattern (BASELINE config 5): terms for the structural non-zeros only, row-major"
struct SparseAffine
    C::Parameter
    x::Vector{Variable}
    d::Union{Nothing, Parameter}
    sign::Int
end

"x (+|-) l for x::Vector{Variable} and a vector Parameter l: one term (1.0, x[i]) per row (src/lazyexpression.jl:249-258 -> vecsubtract!, src/functions.jl:421)"
struct VarBounds
    x::Vector{Variable}
    l::Parameter
    sign::Int
end

"s * (A*x (+|-) b) for a scalar Parameter s: scale!(dest, s, y) (src/lazyexpression.jl:284-290, src/functions.jl:895-915)"
struct ScaledAffine
    s::Parameter
    inner::DenseAffine
end

"one piece of a constraint function: what `analyse_piece` understands"
const Piece = Union{DenseAffine, VarBounds, ScaledAffine, SparseAffine}

is_sparse_param(a) = a isa Parameter && a() isa SparseMatrixCSC{Float64, Int64}

function analyse_sparse(e::LazyExpression)::Union{Nothing, SparseAffine}
    if e.f === Functions.matvecmul! && length(e.args) == 3 && e.args[3] isa Vector{Variable} && is_sparse_param(e.args[2])
        return SparseAffine(e.args[2], e.args[3], nothing, 0)
    elseif (e.f === Functions.vecsubtract! || e.f === Functions.vecadd!) && length(e.args) == 3 && e.args[3] isa Parameter
        inner = unwrap(e.args[2])
        if inner isa LazyExpression
            sp = analyse_sparse(inner)
            sp !== nothing && sp.d === nothing && return SparseAffine(sp.C, sp.x, e.args[3], e.f === Functions.vecsubtract! ? -1 : 1)
        end
    end
    nothing
end

function analyse_piece(arg)::Piece
    e = unwrap(arg)
    e isa LazyExpression || throw(Unsupported("affine operand $(typeof(e))"))
    if (e.f === Functions.vecsubtract! || e.f === Functions.vecadd!) && length(e.args) == 3 && e.args[2] isa Vector{Variable} && e.args[3] isa Parameter
        return VarBounds(e.args[2], e.args[3], e.f === Functions.vecsubtract! ? -1 : 1)
    elseif e.f === Functions.scale! && length(e.args) == 3 && e.args[2] isa Parameter && e.args[2]() isa Number
        return ScaledAffine(e.args[2], analyse_affine(e.args[3]))
    end
    sp = analyse_sparse(e)
    sp !== nothing && return sp
    analyse_affine(e)
end

"the pieces of a constraint expression: vcat!(dest, pieces...) (src/lazyexpression.jl:276-278, src/functions.jl:969-994) or a single piece"
function analyse_pieces(arg)::Vector{Piece}
    e = unwrap(arg)
    if e isa LazyExpression && e.f === Functions.vcat! && length(e.args) >= 2
        return Piece[analyse_piece(a) for a in e.args[2:end]]
    end
    Piece[analyse_piece(e)]
end

"transpose(x) * Q * y (src/lazyexpression.jl:219-226): bilinearmul!(dest, Q, x', y) with Q a matrix Parameter"
struct Bilinear
    Q::Parameter
    x::Vector{Variable}
    y::Vector{Variable}
end

function analyse_bilinear(expr)::Bilinear
    e = unwrap(expr)
    (e isa LazyExpression && e.f === Functions.bilinearmul! && length(e.args) == 4 && e.args[2] isa Parameter) || throw(Unsupported("not a bilinear objective"))
    xt, y = e.args[3], e.args[4]
    x = xt isa Union{Transpose, Adjoint} ? parent(xt) : throw(Unsupported("bilinearmul! without a transposed left vector"))
    (x isa Vector{Variable} && y isa Vector{Variable}) || throw(Unsupported("bilinearmul! of non-Variable vectors"))
    Bilinear(e.args[2], x, y)
end

"residual ⋅ residual (vecdot!, src/lazyexpression.jl:228-232) or residual' * residual (matvecmul! into a QuadraticFunction, src/functions.jl:824-829)"
function analyse_lsq(expr)::DenseAffine
    e = unwrap(expr)
    e isa LazyExpression || throw(Unsupported("objective $(typeof(e))"))
    if e.f === Functions.vecdot! && length(e.args) == 3
        l, r = unwrap(e.args[2]), unwrap(e.args[3])
        l === r || throw(Unsupported("vecdot! of two different vectors"))
        return analyse_affine(l)
    elseif e.f === Functions.matvecmul! && length(e.args) == 3 && e.args[1] isa Parametron.QuadraticFunction
        r = unwrap(e.args[3])
        lt = unwrap(e.args[2])                      # LazyExpression(adjoint, resid) — the generic rule (:198)
        (lt isa LazyExpression && lt.f === adjoint && unwrap(lt.args[1]) === r) || throw(Unsupported("x' * y with x != y"))
        return analyse_affine(r)
    end
    throw(Unsupported("objective builder $(e.f)"))
end

# ---------------------------------------------------------------------------------------------------------------------------
# device records (↔ Objective / Constraint, src/moi_interop.jl:113-175)

mutable struct HIPObjective
    objective              # the reference's Objective (its .f is the host MOI function the optimizer is given)
    quad::DevPtr           # MOI.ScalarQuadraticTerm[] on the device (C_NULL in the host_csc hand-off: P's CSC values instead)
    lin::DevPtr            # MOI.ScalarAffineTerm[]
    constant::DevPtr       # Float64[1]
    nquad::Int
    nlin::Int
    cbuf::Vector{Float64}  # where the constant is fetched to: preallocated (update! allocates nothing)
    P_values::DevPtr       # host_csc: alpha * P in CSC order, upper triangle
    host_quad::Vector{Float64}   # :moi with a canonical objective: the page-locked memory the objective's quadratic_terms vector views (empty otherwise)
end

struct HIPConstraint
    constraint             # the reference's Constraint
    terms::DevPtr          # MOI.VectorAffineTerm[]
    constants::DevPtr      # Float64[rows]
    nterms::Int
    rows::Int
    dense::Union{Nothing, Tuple{DevPtr, Int, Int, Int, Vector{Float64}}}     # (C buffer, lda, rows, cols, the Parameter's page-locked host copy) when the function is ONE unscaled dense block
end

"what a host solver's update takes, in page-locked arrays the device fills (handoff = :host_csc)"
mutable struct HostQP
    Px::Vector{Float64}    # P, upper triangle, CSC values (structure: column k holds rows 1..k of the variables in index order)
    Ax::Vector{Float64}    # A, CSC values, the constraint blocks stacked in the reference's update order
    small::Vector{Float64} # q | l | u, one transfer
    n::Int
    m::Int
    small_dev::DevPtr
end
q_of(h::HostQP) = view(h.small, 1:h.n)
l_of(h::HostQP) = view(h.small, h.n + 1:h.n + h.m)
u_of(h::HostQP) = view(h.small, h.n + h.m + 1:h.n + 2 * h.m)

mutable struct HIPModel
    model::Model
    plan::H.Plan
    params::Vector{DeviceParameter}
    varmap::DevPtr
    objective::Union{Nothing, HIPObjective}
    constraints::Vector{HIPConstraint}
    cpu_records::Vector{Any}          # records that stay on the reference's own update! (unsupported shapes, constant expressions)
    literal_limit::Int                # literal (uncombined) objective up to this many quadratic terms, canonical beyond (DESIGN.md §3)
    slot::Int                         # staging slot of the next update (alternates: the copy of update k+1 does not wait for update k's commits)
    handoff::Symbol                   # :moi (the reference's boundary) or :host_csc
    host::Union{Nothing, HostQP}
    solver_update::Any                # f(Px, Ax, q, l, u), called by update! in the host_csc hand-off
    strict::Bool                      # true: every non-constant record runs on the device (HIPModel throws otherwise); false: unknown shapes keep the CPU update!
    # SMALL models (launch-bound on the device: README Example 1 .. a few hundred variables): Parameter values through mailboxes, the MOI
    # functions' own vectors registered as the kernels' outputs, update! = ONE library call (INTEGRATION.md section 5)
    small::Bool
    run::Union{Nothing, H.ModelRun}
    registered::Vector{Any}           # the host arrays the kernels store into (kept alive and un-resized for the life of the plan)
end

"""on_device(hm) -> (objective = :device | :cpu | :constant, constraints = [:device | :cpu | :constant ...] in the reference's update order).
A record is :constant when the reference never updates it (src/moi_interop.jl:132,169), :cpu only under `strict = false`."""
function on_device(hm::HIPModel)
    where_is(r) = r.isconstant ? :constant : (any(x -> x === r, hm.cpu_records) ? :cpu : :device)
    (objective = where_is(hm.model.objective), constraints = Symbol[where_is(c) for c in constraint_records(hm.model)])
end

unsupported_record(what::AbstractString, reason) =
    ArgumentError("ParametronHIP: $what is not a shape the device path knows ($reason). HIPModel(model; strict = false) keeps the reference's CPU update! for it.")

function device_param!(hm::HIPModel, p::Parameter)
    for d in hm.params
        d.param === p && return d
    end
    d = DeviceParameter(hm.plan, p)
    push!(hm.params, d)
    d
end

function upload_indices(plan::H.Plan, v::Vector{Int64})
    buf = H.alloc(plan, 8 * max(length(v), 1))
    H.upload!(plan, buf, v)
    buf
end

"(A buffer, lda, rows, cols) of the operand a node reads; A' * x reads the device transposition of A (pmt_transpose_f64, recorded on the tape)"
function operand!(hm::HIPModel, da::DenseAffine, rec)
    d = device_param!(hm, da.A)
    da.transposed || return d.buf, d.ld, d.rows, d.cols
    rows, cols = d.cols, d.rows
    ld = padded_ld(rows)
    t = H.alloc(hm.plan, 8 * ld * max(cols, 1))
    H.transpose!(t, ld, d.buf, d.ld, d.rows, d.cols, rec)
    t, ld, rows, cols
end

# ---- small models
const SMALL_MODEL_ELEMENTS = 262144          # host-updated Parameter elements up to which the small-model path wins (DESIGN.md section 4: measured crossover)

dense_f64(v) = v isa Matrix{Float64} || v isa Vector{Float64} || v isa Float64
"the reference's boundary (:moi), every Parameter a dense Float64 array / number, and few enough elements that update! is launch-bound"
function small_model(model::Model, handoff::Symbol)
    handoff === :moi || return false
    total = 0
    for p in model.params
        v = p()
        dense_f64(v) || return false
        total += length(v)
    end
    total <= SMALL_MODEL_ELEMENTS
end

"""An MOI buffer of a record.  In a small model: a page-locked vector (H.host_alloc_as) that BECOMES the function object's vector — the kernel
stores into it from inside the one launch and there is no device twin; returns (device-visible pointer, the vector).  Otherwise a plan-owned
device buffer that update! fetches; returns (pointer, nothing)."""
function output_buffer(hm::HIPModel, ::Type{T}, n::Integer) where {T}
    (hm.small && n > 0) || return H.alloc(hm.plan, sizeof(T) * max(n, 1)), nothing
    v = H.host_alloc_as(T, n)
    push!(hm.registered, v)                    # (kept alive — and never resized — for the life of the plan)
    DevPtr(pointer(v)), v
end
"""the objective's MOI buffers: in a small model its quadratic_terms / affine_terms become page-locked vectors the kernels store into (the
function object is mutable, src/moi_interop.jl:44-53, as the delivery path below already relies on)"""
function adopt_objective_buffers!(hm::HIPModel, objective, nq::Integer, nl::Integer)
    quad, qv = output_buffer(hm, MOI.ScalarQuadraticTerm{Float64}, nq)
    lin, lv = output_buffer(hm, MOI.ScalarAffineTerm{Float64}, nl)
    qv === nothing || (objective.f.quadratic_terms = qv)
    lv === nothing || (objective.f.affine_terms = lv)
    quad, lin
end
"the scalar functions' constant: a page-locked word the kernel stores into (small models) or a device word that is fetched"
function constant_buffer(hm::HIPModel)
    hm.small || return H.alloc(hm.plan, 8), zeros(1)
    word = H.host_alloc(1)
    DevPtr(pointer(word)), word
end

"front of a small model's tape: one mailbox per Parameter and the entry that copies it into the Parameter's device buffer"
function record_mailboxes!(hm::HIPModel, rec)
    for p in hm.model.params
        d = device_param!(hm, p)
        n = d.cols > 0 ? d.ld * d.cols : max(d.ld, 1)
        d.mailbox = H.host_alloc(n)            # zero filled: the padding rows stay zero
        H.copy_bytes!(d.buf, DevPtr(pointer(d.mailbox)), 8 * n, rec)
    end
    nothing
end

"after the tape is recorded: the run that update! calls — every Parameter's own value array registered with its strides"
function create_run!(hm::HIPModel)
    run = H.ModelRun(hm.plan)
    for d in hm.params
        v = d.param()
        if d.cols > 0
            d.host_ptr = pointer(v)
            d.slot = H.add_mailbox!(run, d.host_ptr, d.rows, d.cols, 1, d.rows, d.mailbox, d.ld)
        elseif d.cols == 0
            d.host_ptr = pointer(v)
            d.slot = H.add_mailbox!(run, d.host_ptr, d.rows, 0, 1, 0, d.mailbox, max(d.ld, 1))
        else
            d.scalar = [Float64(v)]
            d.host_ptr = pointer(d.scalar)
            d.slot = H.add_mailbox!(run, d.host_ptr, 1, 0, 1, 0, d.mailbox, 1)
        end
    end
    hm.run = run
    nothing
end

identity_map(hm::HIPModel, x::Vector{Variable}) = all(i -> hm.model.model_var_to_optimizer[x[i].index].value == i, eachindex(x)) && length(x) == length(hm.model.model_var_to_optimizer)

function record_objective!(hm::HIPModel, objective, rec)
    sense_sign = hm.model.backend.sense == MOI.MAX_SENSE ? -1.0 : 1.0          # a host QP solver minimises (DESIGN.md §8)
    bil = try analyse_bilinear(objective.expr) catch err; err isa Unsupported ? nothing : rethrow() end
    if bil !== nothing
        # transpose(x) * Q * y: quad[k] = (Q[k], x[(k-1) ÷ n + 1], y[(k-1) mod n + 1]) in column-major order of Q (src/functions.jl:840-858)
        hm.handoff === :host_csc && throw(Unsupported("bilinear objective in the host_csc hand-off"))
        d = device_param!(hm, bil.Q)
        xv = upload_indices(hm.plan, Int64[v.index for v in bil.x]); yv = upload_indices(hm.plan, Int64[v.index for v in bil.y])
        nq = d.rows * d.cols
        quad, lin = adopt_objective_buffers!(hm, objective, nq, 0)
        constant, cbuf = constant_buffer(hm)
        H.bilinear!(quad, d.buf, d.ld, d.rows, d.cols, xv, yv, 1, hm.varmap, rec)
        return HIPObjective(objective, quad, lin, constant, nq, 0, cbuf, DevPtr(C_NULL), Float64[])
    end
    da = analyse_lsq(objective.expr)
    A, lda, r, n = operand!(hm, da, rec)
    xvar = upload_indices(hm.plan, Int64[v.index for v in da.x])
    b = da.b === nothing ? DevPtr(C_NULL) : device_param!(hm, da.b).buf
    if hm.handoff === :host_csc
        identity_map(hm, da.x) || throw(Unsupported("host_csc hand-off needs the optimizer to keep the variable order"))
        nq = div(n * (n + 1), 2)
        Pv, lin, constant = H.alloc(hm.plan, 8 * nq), H.alloc(hm.plan, 16 * n), H.alloc(hm.plan, 8)
        ws = H.alloc(hm.plan, H.quad_gram_workspace_bytes(r, n))
        hm.host.Px = H.host_alloc(nq)
        H.quad_gram_csc_deliver!(Pv, hm.host.Px, lin, constant, A, lda, padded_rows(r), n, xvar, b, da.sign, hm.varmap, sense_sign, ws, rec)
        return HIPObjective(objective, DevPtr(C_NULL), lin, constant, 0, n, zeros(1), Pv, Float64[])
    end
    if r * n * n <= hm.literal_limit
        # literal: the reference's term order and coefficients bit for bit (src/functions.jl:702-709 over :548-576, moi_interop.jl:45-62)
        nq, nl = r * n * n, 2 * r * n
        res, resc = H.alloc(hm.plan, 16 * r * n), H.alloc(hm.plan, 8 * r)
        quad, lin = adopt_objective_buffers!(hm, objective, nq, nl)
        constant, cbuf = constant_buffer(hm)
        H.affine_assemble!(res, resc, A, lda, r, n, xvar, b, da.sign, rec)
        H.quad_expand!(quad, lin, constant, r, res, n, resc, res, n, resc, 1, hm.varmap, rec)
        return HIPObjective(objective, quad, lin, constant, nq, nl, cbuf, DevPtr(C_NULL), Float64[])
    end
    issorted([v.index for v in da.x], lt = <=) || throw(Unsupported("canonical objective needs strictly increasing variables"))
    nq = div(n * (n + 1), 2)
    if hm.small
        # a small model: the node stores its terms straight into the function object's own (registered) vectors; nothing to deliver
        quad, lin = adopt_objective_buffers!(hm, objective, nq, n)
        constant, cbuf = constant_buffer(hm)
        ws = H.alloc(hm.plan, H.quad_gram_workspace_bytes(padded_rows(r), n))
        H.quad_gram!(quad, lin, constant, A, lda, padded_rows(r), n, xvar, b, da.sign, 1, hm.varmap, ws, rec)
        return HIPObjective(objective, quad, lin, constant, nq, n, cbuf, DevPtr(C_NULL), Float64[])
    end
    quad, lin, constant = H.alloc(hm.plan, 24 * nq), H.alloc(hm.plan, 16 * n), H.alloc(hm.plan, 8)
    ws = H.alloc(hm.plan, H.quad_gram_workspace_bytes(r, n))
    # The reference's own boundary, overlapped: the objective's quadratic_terms vector is made a view of page-locked memory (24-byte isbits
    # MOI.ScalarQuadraticTerm{Float64} = pmt_quadratic_term; the function object is mutable, src/moi_interop.jl:44-47) and the contraction
    # delivers the terms into it row band by row band while it runs (pmt_quad_gram_deliver_f64): nothing is copied on the host afterwards.
    T = MOI.ScalarQuadraticTerm{Float64}
    if isbitstype(T) && sizeof(T) == 24 && nq > 0
        hq = H.host_alloc(3 * nq)
        H.quad_gram_deliver!(quad, Ptr{Cvoid}(pointer(hq)), lin, constant, A, lda, padded_rows(r), n, xvar, b, da.sign, 1, hm.varmap, ws, rec)
        objective.f.quadratic_terms = unsafe_wrap(Vector{T}, Ptr{T}(pointer(hq)), nq)
        return HIPObjective(objective, quad, lin, constant, nq, n, zeros(1), DevPtr(C_NULL), hq)
    end
    H.quad_gram!(quad, lin, constant, A, lda, padded_rows(r), n, xvar, b, da.sign, 1, hm.varmap, ws, rec)
    HIPObjective(objective, quad, lin, constant, nq, n, zeros(1), DevPtr(C_NULL), Float64[])
end

piece_rows(hm::HIPModel, p::DenseAffine) = (d = device_param!(hm, p.A); p.transposed ? d.cols : d.rows)
piece_rows(hm::HIPModel, p::VarBounds) = length(p.x)
piece_rows(hm::HIPModel, p::SparseAffine) = device_param!(hm, p.C).rows
piece_terms(hm::HIPModel, p::SparseAffine) = device_param!(hm, p.C).nnz
piece_rows(hm::HIPModel, p::ScaledAffine) = piece_rows(hm, p.inner)
piece_terms(hm::HIPModel, p::DenseAffine) = (d = device_param!(hm, p.A); d.rows * d.cols)
piece_terms(hm::HIPModel, p::VarBounds) = length(p.x)
piece_terms(hm::HIPModel, p::ScaledAffine) = piece_terms(hm, p.inner)

"one piece of a constraint function -> MOI.VectorAffineTerms at term offset `t0`, rows from `row0` (vcat!: the pieces are stacked)"
function record_piece!(hm::HIPModel, p::DenseAffine, terms::DevPtr, constants::DevPtr, t0::Int, row0::Int, rec)
    A, lda, r, n = operand!(hm, p, rec)
    xvar = upload_indices(hm.plan, Int64[v.index for v in p.x])
    b = p.b === nothing ? DevPtr(C_NULL) : device_param!(hm, p.b).buf
    H.affine_pack_vector!(terms + 24 * t0, constants + 8 * row0, A, lda, r, n, xvar, b, p.sign, hm.varmap, row0, rec)
end
"""sparse block: the row-major order of the pattern is worked out once (pmt_sparse_rowmajor_order); per re-evaluation one launch scatters nzval into
the MOI terms — the block form (CSC -> row-major through LDS, 36 bytes per non-zero) where the pattern allows it and a row's share of a column
band is long enough to be written as a run, the slab form (one column slab per XCD) otherwise; constants 0 (+|-) d by pmt_consts_f64"""
function record_piece!(hm::HIPModel, p::SparseAffine, terms::DevPtr, constants::DevPtr, t0::Int, row0::Int, rec)
    d = device_param!(hm, p.C)
    C = p.C()
    plan = H.sparse_plan(C)
    xmap = Int64[hm.model.model_var_to_optimizer[v.index].value for v in p.x]          # the optimizer's index of every column's variable
    blk = H.sparse_block_plan(C, plan)
    ncb = blk.cw > 0 ? cld(d.cols, blk.cw) : 0
    if blk.cw > 0 && d.nnz >= 16 * d.rows * ncb
        desc, idx, band = H.alloc(hm.plan, 8 * length(blk.desc)), H.alloc(hm.plan, 4 * max(length(blk.idx), 1)), H.alloc(hm.plan, 8 * length(blk.band_ptr))
        H.upload!(hm.plan, desc, blk.desc); H.upload!(hm.plan, idx, blk.idx); H.upload!(hm.plan, band, blk.band_ptr)
        dd = p.d === nothing ? DevPtr(C_NULL) : device_param!(hm, p.d).buf
        H.sparse_pack_vector_blocks!(terms + 24 * t0, constants + 8 * row0, d.buf, desc, idx, band, upload_indices(hm.plan, xmap), d.rows, d.cols, d.nnz, blk.cw,
                                     DevPtr(C_NULL), row0, dd, p.d === nothing ? 0 : p.sign, rec)
    else
        perm, slab = upload_indices(hm.plan, plan.perm), upload_indices(hm.plan, plan.slab_ptr)
        tvar = upload_indices(hm.plan, Int64[xmap[c] for c in plan.term_col[1:max(d.nnz, 0)]])
        H.sparse_pack_vector!(terms + 24 * t0, d.buf, perm, tvar, slab, d.rows, 8, DevPtr(C_NULL), row0, rec)
        # without a d the constants keep the zeros the plan's allocation was filled with
        p.d === nothing || H.consts!(constants + 8 * row0, device_param!(hm, p.d).buf, d.rows, p.sign, rec)
    end
end
function record_piece!(hm::HIPModel, p::VarBounds, terms::DevPtr, constants::DevPtr, t0::Int, row0::Int, rec)
    xvar = upload_indices(hm.plan, Int64[v.index for v in p.x])
    H.vars_addsub!(DevPtr(C_NULL), terms + 24 * t0, constants + 8 * row0, xvar, length(p.x), device_param!(hm, p.l).buf, p.sign, hm.varmap, row0, rec)
end
function record_piece!(hm::HIPModel, p::ScaledAffine, terms::DevPtr, constants::DevPtr, t0::Int, row0::Int, rec)
    # the native block, scaled by the device scalar, then the MOI copy (scale! works on materialised terms: src/functions.jl:895-915)
    A, lda, r, n = operand!(hm, p.inner, rec)
    xvar = upload_indices(hm.plan, Int64[v.index for v in p.inner.x])
    b = p.inner.b === nothing ? DevPtr(C_NULL) : device_param!(hm, p.inner.b).buf
    res, resc, sc = H.alloc(hm.plan, 16 * r * n), H.alloc(hm.plan, 8 * r), H.alloc(hm.plan, 16 * r * n)
    H.affine_assemble!(res, resc, A, lda, r, n, xvar, b, p.inner.sign, rec)
    H.affvec_scale!(sc, constants + 8 * row0, r, r * n, res, resc, device_param!(hm, p.s).buf, rec)
    H.pack_vector_affine!(terms + 24 * t0, sc, r, n, hm.varmap, row0, rec)
end

function record_constraint!(hm::HIPModel, constraint, rec)
    pieces = analyse_pieces(constraint.expr)
    rows = sum(p -> piece_rows(hm, p), pieces)
    nterms = sum(p -> piece_terms(hm, p), pieces)
    terms, tv = output_buffer(hm, MOI.VectorAffineTerm{Float64}, nterms)
    constants, cv = output_buffer(hm, Float64, rows)
    if tv !== nothing && cv !== nothing
        # MOI.VectorAffineFunction is immutable, the reference's Constraint record is not (src/moi_interop.jl:141-147): the record gets a function
        # object of the same type whose vectors ARE the page-locked ones
        constraint.f = typeof(constraint.f)(tv, cv)
    end
    # A constraint that reads Parameter values only (no node of the tape feeds it) is independent of every other record (update! of one
    # Constraint, src/moi_interop.jl:168-175): beside a canonical least-squares objective it goes to the plan's side lane
    independent = all(p -> !(p isa DenseAffine && p.transposed) && !(p isa ScaledAffine), pieces)
    # (never in a small model: its Parameter values arrive through entries at the FRONT of the tape, which side-lane entries would overtake)
    side = !hm.small && independent && hm.objective !== nothing && hm.objective.nlin > 0 && (hm.objective.nquad == 0 || hm.objective.nquad == div(hm.objective.nlin * (hm.objective.nlin + 1), 2))
    side && H.set_lane!(hm.plan, 1)
    t0 = 0; row0 = 0
    for p in pieces
        record_piece!(hm, p, terms, constants, t0, row0, rec)
        t0 += piece_terms(hm, p); row0 += piece_rows(hm, p)
    end
    side && H.set_lane!(hm.plan, 0)
    dense = nothing
    if length(pieces) == 1 && pieces[1] isa DenseAffine && !pieces[1].transposed
        d = device_param!(hm, pieces[1].A)
        dense = (d.buf, d.ld, d.rows, d.cols, d.host)
    end
    HIPConstraint(constraint, terms, constants, nterms, rows, dense)
end

"all Constraint records of the model in the reference's update order (the fields of Parametron.Constraints, src/moi_interop.jl:195-262)"
function constraint_records(model::Model)
    out = Any[]
    cs = model.constraints
    for name in fieldnames(typeof(cs))
        append!(out, getfield(cs, name))
    end
    out
end

set_kind(::MOI.Zeros) = 0
set_kind(::MOI.Nonnegatives) = 1
set_kind(::MOI.Nonpositives) = 2

"""host_csc: q, l, u on the device (one block) and the recorded fetches.  A's CSC values ARE the dense blocks' Parameter values column by
column (stacked), so they leave as pitched copies of the Parameter buffers — no kernel at all; q is the coefficient field of the objective's
affine terms, l / u come from the constraint constants and the cone (pmt_qp_bounds_f64)."""
function record_host_handoff!(hm::HIPModel, rec)
    h = hm.host
    n, m = h.n, h.m
    h.small = H.host_alloc(n + 2 * m)
    h.small_dev = H.alloc(hm.plan, 8 * (n + 2 * m))
    h.Ax = H.host_alloc(n * m)
    o = hm.objective
    ident = upload_indices(hm.plan, Int64[i - 1 for i in 1:n])
    seg = upload_indices(hm.plan, Int64[i - 1 for i in 1:n + 1])
    H.set_lane!(hm.plan, 1)                         # behind the Gram node's affine part on the side stream (plan.hip `replay`)
    H.csc_values!(h.small_dev, o.lin, 16, n, ident, seg, n, hm.model.backend.sense == MOI.MAX_SENSE ? -1.0 : 1.0, DevPtr(C_NULL), rec)
    row0 = 0
    for c in hm.constraints
        H.qp_bounds!(h.small_dev + 8 * (n + row0), h.small_dev + 8 * (n + m + row0), c.constants, c.rows, set_kind(c.constraint.set), 0.0, 1e20, rec)
        row0 += c.rows
    end
    H.record_fetch!(hm.plan, h.small, h.small_dev)
    # A's values are NOT fetched: every Parameter of this host is evaluated on the host (refresh! leaves its value in the page-locked staging
    # array d.host), so a dense block's CSC values are already here — copy_A! copies them on the host while the device works.  (A host whose
    # Parameter values live on the device records pitched fetches out of the Parameter buffers instead: pmt_plan_record_fetch_2d at the front
    # of the side lane, ParametronHIP.record_fetch_matrix!; the Python host of this repository does both.)
    H.set_lane!(hm.plan, 0)
    nothing
end

"A's values: one host-side pitched copy per dense block, from the Parameter's page-locked host copy into the block's row range of every column
(column j of A = the blocks' columns j, stacked) on a few worker threads — pmt_host_copy_2d — while the device re-evaluates the objective"
function copy_A!(hm::HIPModel)
    h = hm.host
    row0 = 0
    for c in hm.constraints
        buf, lda, r, n, src = c.dense
        H.host_copy_matrix!(pointer(h.Ax) + 8 * row0, 8 * h.m, pointer(src), 8 * r, r, n)
        row0 += r
    end
    nothing
end

"A's values out of the DEVICE copies instead, behind the re-evaluation (kept for hosts whose Parameter values are not on the host): one pitched copy per
dense block straight out of its Parameter buffer"
function fetch_A!(hm::HIPModel)
    h = hm.host
    row0 = 0
    for c in hm.constraints
        buf, lda, r, n, _ = c.dense
        H.fetch_matrix!(hm.plan, pointer(h.Ax) + 8 * row0, 8 * h.m, buf, lda, r, n)
        row0 += r
    end
    nothing
end

function HIPModel(model::Model; device::Integer = 0, literal_limit::Integer = 1 << 24, handoff::Symbol = :moi, solver_update = nothing, strict::Bool = true)
    handoff in (:moi, :host_csc) || throw(ArgumentError("handoff must be :moi or :host_csc"))
    model.initialized || Parametron.initialize!(model)              # copy_to + mapindices! first: the index map is then final (src/model.jl:117-122)
    plan = H.Plan(device)
    nvars = length(model.model_var_to_optimizer)
    varmap = upload_indices(plan, Int64[vi.value for vi in model.model_var_to_optimizer])          # src/model.jl:100-107
    host = handoff === :host_csc ? HostQP(Float64[], Float64[], Float64[], nvars, 0, DevPtr(C_NULL)) : nothing
    small = small_model(model, handoff)
    hm = HIPModel(model, plan, DeviceParameter[], varmap, nothing, HIPConstraint[], Any[], literal_limit, 0, handoff, host, solver_update, strict, small, nothing, Any[])
    rec = H.recording_stream(plan)
    H.begin_record!(plan)
    try
        small && record_mailboxes!(hm, rec)
        obj = model.objective
        if obj.isconstant
            nothing                                                  # never updated (src/moi_interop.jl:132)
        else
            try
                hm.objective = record_objective!(hm, obj, rec)
            catch err
                err isa Unsupported || rethrow()
                handoff === :host_csc && rethrow()                   # the host_csc hand-off has no CPU fallback for a record: say so
                strict && throw(unsupported_record("the objective", err.what))
                @info "ParametronHIP: the objective stays on the CPU path (strict = false)" reason = err.what
                push!(hm.cpu_records, obj)
            end
        end
        for c in constraint_records(model)
            c.isconstant && continue                                 # src/moi_interop.jl:169
            try
                push!(hm.constraints, record_constraint!(hm, c, rec))
            catch err
                err isa Unsupported || rethrow()
                handoff === :host_csc && rethrow()
                strict && throw(unsupported_record("constraint $(length(hm.constraints) + length(hm.cpu_records) + 1) ($(typeof(c.set)))", err.what))
                @info "ParametronHIP: a constraint stays on the CPU path (strict = false)" reason = err.what
                push!(hm.cpu_records, c)
            end
        end
        if handoff === :host_csc
            (hm.objective !== nothing && all(c -> c.dense !== nothing, hm.constraints)) ||
                throw(ArgumentError("the host_csc hand-off covers a least-squares objective and unscaled dense constraint blocks"))
            host.m = sum(c -> c.rows, hm.constraints; init = 0)
            record_host_handoff!(hm, rec)
        end
    finally
        H.end_record!(plan)
    end
    small && create_run!(hm)
    hm
end

# ---------------------------------------------------------------------------------------------------------------------------
# update! / solve! (src/model.jl:132-159)

"update!(objective, optimizer, varmap) of src/moi_interop.jl:131-137 with the builders and the MOI copy done on the device"
function Parametron.update!(o::HIPObjective, hm::HIPModel, optimizer)
    f = o.objective.f
    resize!(f.affine_terms, o.nlin)                                  # in place, as the reference does (:48,53): no-ops after the first solve
    if isempty(o.host_quad)
        resize!(f.quadratic_terms, o.nquad)
        H.fetch!(hm.plan, f.quadratic_terms, o.quad)
    else
        H.fetch_synchronize(hm.plan)                                 # the delivered quadratic terms have landed in f.quadratic_terms' memory
    end
    H.fetch!(hm.plan, f.affine_terms, o.lin)
    H.fetch!(hm.plan, o.cbuf, o.constant)
    H.synchronize(hm.plan)
    f.constant = o.cbuf[1]
    MOI.set(optimizer, MOI.ObjectiveFunction{typeof(f)}(), f)
    nothing
end

"update!(constraint, optimizer, varmap) of src/moi_interop.jl:168-175"
function Parametron.update!(c::HIPConstraint, hm::HIPModel, optimizer)
    f = c.constraint.f
    resize!(f.terms, c.nterms)
    resize!(f.constants, c.rows)
    H.fetch!(hm.plan, f.terms, c.terms)
    H.fetch!(hm.plan, f.constants, c.constants)
    H.synchronize(hm.plan)
    MOI.set(optimizer, MOI.ConstraintFunction(), c.constraint.optimizerindex, f)
    nothing
end

"""update!(m::Model) of src/model.jl:132-143: setdirty!, Parameters, one tape replay, then the hand-off.  Nothing below allocates: the
Parameter values go through their page-locked staging arrays onto the copy stream, the commits and the tape onto the plan's stream."""
"""update!(m::Model) of a SMALL model: the callbacks here (they are Julia functions), everything else — value -> mailbox, the one launch, the
wait — in pmt_model_update; the kernels have stored the MOI terms into the function objects' own vectors when it returns.  Allocates nothing."""
function update_small!(hm::HIPModel)
    m = hm.model
    setdirty!(m)
    for d in hm.params
        v = d.param()                                                # evaluates the callback (src/parameter.jl:93-99)
        if d.cols == -1
            d.scalar[1] = v
        elseif pointer(v) != d.host_ptr                              # an out-of-place callback returned a new array
            d.host_ptr = pointer(v)
            H.set_host!(hm.run, d.slot, d.host_ptr)
        end
    end
    H.model_update!(hm.run, nothing, true)                           # every slot is dirty: setdirty!(model) (src/model.jl:132-133)
    o = hm.objective
    if o !== nothing
        f = o.objective.f
        f.constant = o.cbuf[1]
        MOI.set(m.optimizer, MOI.ObjectiveFunction{typeof(f)}(), f)  # src/moi_interop.jl:134
    end
    for c in hm.constraints
        MOI.set(m.optimizer, MOI.ConstraintFunction(), c.constraint.optimizerindex, c.constraint.f)      # src/moi_interop.jl:171
    end
    hm.strict || cpu_update!(hm)
    nothing
end

function Parametron.update!(hm::HIPModel)
    hm.small && return update_small!(hm)
    m = hm.model
    setdirty!(m)
    slot = hm.slot
    hm.slot = 1 - slot
    H.stage_slot!(hm.plan, slot)
    for d in hm.params
        refresh!(hm.plan, d, slot)                                   # runs the user's callback once (dirty flag) and starts the upload
    end
    for d in hm.params
        commit!(hm.plan, d, slot)                                    # plan stream: staging -> the buffer the kernels read
    end
    H.staging_consumed!(hm.plan)
    H.update!(hm.plan)                                               # every device node of the model (+ the recorded fetches / the delivery of P)
    if hm.handoff === :host_csc
        h = hm.host
        copy_A!(hm)                                                  # A's blocks: host -> host, while the device works (no allocation: worker threads of the library)
        H.fetch_synchronize(hm.plan)                                 # q | l | u and P's band groups have landed
        H.synchronize(hm.plan)
        hm.solver_update === nothing || hm.solver_update(h.Px, h.Ax, q_of(h), l_of(h), u_of(h))
        return nothing
    end
    hm.objective === nothing || Parametron.update!(hm.objective, hm, m.optimizer)
    for c in hm.constraints
        Parametron.update!(c, hm, m.optimizer)
    end
    hm.strict || cpu_update!(hm)
    nothing
end

"strict = false only: the records whose shape the device path does not know run the reference's own update! (src/moi_interop.jl:131-137,168-175)"
function cpu_update!(hm::HIPModel)
    m = hm.model
    for r in hm.cpu_records
        Parametron.update!(r, m.optimizer, m.model_var_to_optimizer)
    end
    nothing
end

function solve!(hm::HIPModel)                                        # src/model.jl:151-159
    Parametron.update!(hm)
    MOI.optimize!(hm.model.optimizer)
    nothing
end

end # module
