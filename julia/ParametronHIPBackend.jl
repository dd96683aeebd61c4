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
# Anything else makes `HIPModel` keep the reference's own CPU `update!` for that record (it says so once): the device path is an
# accelerator for the shapes it knows, never a silent approximation.
#
# NOT EXECUTED in this repository's CI (no julia in the build image; tests/test_gpu_julia.py runs julia/example1_parity.jl when a julia
# binary is present on the GPU box).  tests/test_cabi_exports.py checks every ccall of julia/*.jl against include/parametron_hip.h —
# symbol, argument count and argument type classes.
module ParametronHIPBackend

export HIPModel, solve!

import Parametron
import Parametron: Model, Parameter, Variable, LazyExpression, setdirty!
import Parametron.Functions
import MathOptInterface
const MOI = MathOptInterface
using LinearAlgebra

import ..ParametronHIP
const H = ParametronHIP
const DevPtr = H.DevPtr

# ---------------------------------------------------------------------------------------------------------------------------
# device mirrors of Parameters (src/parameter.jl:36-104): the callback stays a host function; its value is uploaded when it ran

mutable struct DeviceParameter
    param::Parameter
    buf::DevPtr
    rows::Int
    cols::Int            # 0 for vectors
    ld::Int              # leading dimension of the device copy (rows rounded up to 16, + 64 when a multiple of 512: DESIGN.md §2)
end

padded_rows(r) = r >= 64 ? 16 * cld(r, 16) : r
padded_ld(r) = (p = padded_rows(r); p >= 512 && p % 512 == 0 ? p + 64 : p)

function DeviceParameter(plan::H.Plan, p::Parameter)
    val = p()
    if val isa AbstractMatrix
        r, c = size(val)
        ld = padded_ld(r)
        buf = H.alloc(plan, 8 * ld * max(c, 1))            # zero filled: the padding rows stay zero
        return DeviceParameter(p, buf, r, c, ld)
    elseif val isa AbstractVector
        r = length(val)
        return DeviceParameter(p, H.alloc(plan, 8 * max(padded_rows(r), 1)), r, 0, padded_rows(r))
    end
    throw(ArgumentError("Parameters of type $(typeof(val)) are not supported on the device"))
end

"evaluate the Parameter (runs its update function if dirty, src/parameter.jl:93-99) and upload the value"
function refresh!(plan::H.Plan, d::DeviceParameter)
    val = d.param()
    if d.cols > 0
        H.upload_matrix!(plan, d.buf, d.ld, Matrix{Float64}(val))
    else
        H.upload!(plan, d.buf, Vector{Float64}(val))
    end
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

struct HIPObjective
    objective              # the reference's Objective (its .f is the host MOI function the optimizer is given)
    quad::DevPtr           # MOI.ScalarQuadraticTerm[] on the device
    lin::DevPtr            # MOI.ScalarAffineTerm[]
    constant::DevPtr       # Float64[1]
    nquad::Int
    nlin::Int
end

struct HIPConstraint
    constraint             # the reference's Constraint
    terms::DevPtr          # MOI.VectorAffineTerm[]
    constants::DevPtr      # Float64[rows]
    nterms::Int
    rows::Int
end

mutable struct HIPModel
    model::Model
    plan::H.Plan
    params::Vector{DeviceParameter}
    varmap::DevPtr
    objective::Union{Nothing, HIPObjective}
    constraints::Vector{HIPConstraint}
    cpu_records::Vector{Any}          # records that stay on the reference's own update! (unsupported shapes, constant expressions)
    literal_limit::Int                # literal (uncombined) objective up to this many quadratic terms, canonical beyond (DESIGN.md §3)
end

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

function record_objective!(hm::HIPModel, objective, rec)
    da = analyse_lsq(objective.expr)
    A, lda, r, n = operand!(hm, da, rec)
    xvar = upload_indices(hm.plan, Int64[v.index for v in da.x])
    b = da.b === nothing ? DevPtr(C_NULL) : device_param!(hm, da.b).buf
    if r * n * n <= hm.literal_limit
        # literal: the reference's term order and coefficients bit for bit (src/functions.jl:702-709 over :548-576, moi_interop.jl:45-62)
        nq, nl = r * n * n, 2 * r * n
        res, resc = H.alloc(hm.plan, 16 * r * n), H.alloc(hm.plan, 8 * r)
        quad, lin, constant = H.alloc(hm.plan, 24 * nq), H.alloc(hm.plan, 16 * nl), H.alloc(hm.plan, 8)
        H.affine_assemble!(res, resc, A, lda, r, n, xvar, b, da.sign, rec)
        H.quad_expand!(quad, lin, constant, r, res, n, resc, res, n, resc, 1, hm.varmap, rec)
        return HIPObjective(objective, quad, lin, constant, nq, nl)
    end
    issorted([v.index for v in da.x], lt = <=) || throw(Unsupported("canonical objective needs strictly increasing variables"))
    nq = div(n * (n + 1), 2)
    quad, lin, constant = H.alloc(hm.plan, 24 * nq), H.alloc(hm.plan, 16 * n), H.alloc(hm.plan, 8)
    ws = H.alloc(hm.plan, H.quad_gram_workspace_bytes(r, n))
    H.quad_gram!(quad, lin, constant, A, lda, padded_rows(r), n, xvar, b, da.sign, 1, hm.varmap, ws, rec)
    HIPObjective(objective, quad, lin, constant, nq, n)
end

function record_constraint!(hm::HIPModel, constraint, rec)
    da = analyse_affine(constraint.expr)
    A, lda, r, n = operand!(hm, da, rec)
    xvar = upload_indices(hm.plan, Int64[v.index for v in da.x])
    b = da.b === nothing ? DevPtr(C_NULL) : device_param!(hm, da.b).buf
    terms, constants = H.alloc(hm.plan, 24 * r * n), H.alloc(hm.plan, 8 * max(r, 1))
    # A constraint that reads Parameter values only (no transposition recorded on the tape for it) is independent of every other record
    # (update! of one Constraint, src/moi_interop.jl:168-175): beside a canonical least-squares objective it goes to the plan's side lane
    side = !da.transposed && hm.objective !== nothing && hm.objective.nquad == div(n * (n + 1), 2) && hm.objective.nlin == n
    side && H.set_lane!(hm.plan, 1)
    H.affine_pack_vector!(terms, constants, A, lda, r, n, xvar, b, da.sign, hm.varmap, 0, rec)
    side && H.set_lane!(hm.plan, 0)
    HIPConstraint(constraint, terms, constants, r * n, r)
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

function HIPModel(model::Model; device::Integer = 0, literal_limit::Integer = 1 << 24)
    model.initialized || Parametron.initialize!(model)              # copy_to + mapindices! first: the index map is then final (src/model.jl:117-122)
    plan = H.Plan(device)
    nvars = length(model.model_var_to_optimizer)
    varmap = upload_indices(plan, Int64[vi.value for vi in model.model_var_to_optimizer])          # src/model.jl:100-107
    hm = HIPModel(model, plan, DeviceParameter[], varmap, nothing, HIPConstraint[], Any[], literal_limit)
    rec = H.recording_stream(plan)
    H.begin_record!(plan)
    try
        obj = model.objective
        if obj.isconstant
            nothing                                                  # never updated (src/moi_interop.jl:132)
        else
            try
                hm.objective = record_objective!(hm, obj, rec)
            catch err
                err isa Unsupported || rethrow()
                @info "ParametronHIP: the objective stays on the CPU path" reason = err.what
                push!(hm.cpu_records, obj)
            end
        end
        for c in constraint_records(model)
            c.isconstant && continue                                 # src/moi_interop.jl:169
            try
                push!(hm.constraints, record_constraint!(hm, c, rec))
            catch err
                err isa Unsupported || rethrow()
                @info "ParametronHIP: a constraint stays on the CPU path" reason = err.what
                push!(hm.cpu_records, c)
            end
        end
    finally
        H.end_record!(plan)
    end
    hm
end

# ---------------------------------------------------------------------------------------------------------------------------
# update! / solve! (src/model.jl:132-159)

"update!(objective, optimizer, varmap) of src/moi_interop.jl:131-137 with the builders and the MOI copy done on the device"
function Parametron.update!(o::HIPObjective, hm::HIPModel, optimizer)
    f = o.objective.f
    resize!(f.quadratic_terms, o.nquad)                              # in place, as the reference does (:48,53)
    resize!(f.affine_terms, o.nlin)
    H.fetch!(hm.plan, f.quadratic_terms, o.quad)
    H.fetch!(hm.plan, f.affine_terms, o.lin)
    c = Vector{Float64}(undef, 1)
    H.fetch!(hm.plan, c, o.constant)
    H.synchronize(hm.plan)
    f.constant = c[1]
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

"update!(m::Model) of src/model.jl:132-143: setdirty!, Parameters, one tape replay, then the MOI hand-off record by record"
function Parametron.update!(hm::HIPModel)
    m = hm.model
    setdirty!(m)
    for d in hm.params
        refresh!(hm.plan, d)                                         # runs the user's callback once (dirty flag) and uploads
    end
    H.update!(hm.plan)                                               # every device node of the model
    hm.objective === nothing || Parametron.update!(hm.objective, hm, m.optimizer)
    for c in hm.constraints
        Parametron.update!(c, hm, m.optimizer)
    end
    for r in hm.cpu_records                                          # shapes the device path does not know: the reference's own update!
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
