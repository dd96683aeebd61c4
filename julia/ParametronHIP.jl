# ParametronHIP.jl — the reference-side binding a Parametron.jl maintainer would add to route the update! hot path
# through libparametron_hip.so (include/parametron_hip.h): the thin ccall layer.  ParametronHIPBackend.jl (next to this file) builds the
# plan from a Parametron.Model and overrides update! / solve!.  The build image has no Julia (SURVEY.md §0.4): tests/test_cabi_exports.py
# checks every ccall against the header statically (symbol, argument count, argument type classes), tests/test_gpu_julia.py runs
# julia/example1_parity.jl when a julia binary exists on the GPU box; the same C ABI is exercised from Python by tests/.
#
# Usage sketch (README Example 1, README.md:23-57 of the reference):
#
#     using Parametron, ParametronHIP
#     model = Model(optimizer); x = [Variable(model) for _ = 1:n]
#     A = DeviceParameter(rand!, zeros(n, n), plan) ...          # host callback + H2D, or device_uniform!(…)
#     hip = HIPObjective(plan, A, x, b)                            # canonical residual ⋅ residual
#     ...
#     ParametronHIP.update!(plan); fetch!(moi_f.quadratic_terms, hip.quad)   # then MOI.set as in src/moi_interop.jl:134
module ParametronHIP

using SparseArrays

const lib = get(ENV, "PARAMETRON_HIP_LIB", "libparametron_hip.so")

# isbits layouts shared with the C structs (SURVEY.md Appendix C): LinearTerm{Float64} / MOI.ScalarAffineTerm{Float64} = 16 B,
# QuadraticTerm{Float64} / MOI.ScalarQuadraticTerm{Float64} = 24 B, MOI.VectorAffineTerm{Float64} = 24 B.
const DevPtr = Ptr{Cvoid}

struct HIPError <: Exception
    code::Cint
    msg::String
end

function check(code::Cint)
    code == 0 && return nothing
    msg = unsafe_string(ccall((:pmt_last_error, lib), Cstring, ()))
    code == 1 && throw(DimensionMismatch(msg))      # src/functions.jl:780-781 and the other @boundscheck sites
    code == 2 && throw(ArgumentError(msg))          # src/lazyexpression.jl:175,185
    code == 4 && error(msg)                         # src/model.jl:50,61,69
    throw(HIPError(code, msg))
end

# ---- plan: device buffers + tape (↔ dest = deepcopy(expr()) and the FunctionWrapper loop)
mutable struct Plan
    handle::Ptr{Cvoid}
    function Plan(device::Integer = 0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pmt_plan_create, lib), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, ref))
        p = new(ref[])
        finalizer(p -> ccall((:pmt_plan_destroy, lib), Cint, (Ptr{Cvoid},), p.handle), p)
        p
    end
end

function alloc(plan::Plan, bytes::Integer)
    ref = Ref{DevPtr}(C_NULL)
    check(ccall((:pmt_plan_alloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{DevPtr}), plan.handle, bytes, ref))
    ref[]
end
upload!(plan::Plan, dst::DevPtr, src::Array) =
    check(ccall((:pmt_plan_upload, lib), Cint, (Ptr{Cvoid}, DevPtr, Ptr{Cvoid}, Csize_t), plan.handle, dst, src, sizeof(src)))
# `dst` may be moi_f.terms / moi_f.quadratic_terms after resize! — isbits element layouts match the device structs
fetch!(plan::Plan, dst::Array, src::DevPtr) =
    check(ccall((:pmt_plan_fetch, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Csize_t), plan.handle, dst, src, sizeof(dst)))
synchronize(plan::Plan) = check(ccall((:pmt_plan_synchronize, lib), Cint, (Ptr{Cvoid},), plan.handle))
"the plan's device-side error state behind a wait the caller made by other means (a timed-out grid barrier of a fused run throws here)"
check_plan(plan::Plan) = check(ccall((:pmt_plan_check, lib), Cint, (Ptr{Cvoid},), plan.handle))
recording_stream(plan::Plan) = ccall((:pmt_plan_recording_stream, lib), Ptr{Cvoid}, (Ptr{Cvoid},), plan.handle)
begin_record!(plan::Plan) = check(ccall((:pmt_plan_begin_record, lib), Cint, (Ptr{Cvoid},), plan.handle))
end_record!(plan::Plan) = check(ccall((:pmt_plan_end_record, lib), Cint, (Ptr{Cvoid},), plan.handle))
"small plans: runs of small tape entries replay as ONE launch (automatic); `fusion!(plan, false)` replays the tape as recorded"
fusion!(plan::Plan, on::Bool) = check(ccall((:pmt_plan_set_fusion, lib), Cint, (Ptr{Cvoid}, Cint), plan.handle, on ? 1 : 0))
"barrier-separated phases over all fused runs (independent nodes share a phase)"
fused_phases(plan::Plan) = Int(ccall((:pmt_plan_fused_phases, lib), Cint, (Ptr{Cvoid},), plan.handle))
fused_workgroups(plan::Plan) = Int(ccall((:pmt_plan_fused_workgroups, lib), Cint, (Ptr{Cvoid},), plan.handle))
"(fused runs, tape entries they replace, launches-or-entries per replay)"
function fused(plan::Plan)
    g = Ref{Cint}(0); n = Ref{Cint}(0); len = Ref{Int64}(0)
    check(ccall((:pmt_plan_fused, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Int64}), plan.handle, g, n, len))
    Int(g[]), Int(n[]), len[]
end
"while recording: lane 1 = the following calls only read Parameter values and are independent of the rest of the tape (side lane), 0 = back"
set_lane!(plan::Plan, lane::Integer) = check(ccall((:pmt_plan_set_lane, lib), Cint, (Ptr{Cvoid}, Cint), plan.handle, lane))
"One update!(model) worth of kernels (src/model.jl:132-143): replays the tape, no allocation."
update!(plan::Plan) = check(ccall((:pmt_plan_update, lib), Cint, (Ptr{Cvoid},), plan.handle))

# ---- builders (each replaces the Parametron.Functions method named in include/parametron_hip.h)

"matvecmul!(y, A, x) fused with vecadd!/vecsubtract!(dest, y, b)  — src/functions.jl:775-798, :751-764"
affine_assemble!(out_terms::DevPtr, out_consts::DevPtr, A::DevPtr, lda, rows, cols, xvar::DevPtr, b::DevPtr, sign, stream) =
    check(ccall((:pmt_affine_assemble_f64, lib), Cint,
                (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, DevPtr, DevPtr, Ptr{Cvoid}),
                A, lda, rows, cols, xvar, b, sign, out_terms, out_consts, stream))

"the same chain + update!(::MOI.VectorAffineFunction, fs, varmap) — src/moi_interop.jl:64-81"
affine_pack_vector!(out_terms::DevPtr, out_consts::DevPtr, A::DevPtr, lda, rows, cols, xvar::DevPtr, b::DevPtr, sign,
                    varmap::DevPtr, row_offset, stream) =
    check(ccall((:pmt_affine_pack_vector_f64, lib), Cint,
                (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, DevPtr, Int64, DevPtr, DevPtr, Ptr{Cvoid}),
                A, lda, rows, cols, xvar, b, sign, varmap, row_offset, out_terms, out_consts, stream))

"_vecdot!(dest::QuadraticFunction, x, y) literal expansion (+ MOI copy when moi != 0) — src/functions.jl:702-709, :548-576"
quad_expand!(out_quad, out_lin, out_const, rows, xt, nx, xc, yt, ny, yc, moi, varmap, stream) =
    check(ccall((:pmt_quad_expand_f64, lib), Cint,
                (Int64, DevPtr, Int64, DevPtr, DevPtr, Int64, DevPtr, Cint, DevPtr, DevPtr, DevPtr, DevPtr, Ptr{Cvoid}),
                rows, xt, nx, xc, yt, ny, yc, moi, varmap, out_quad, out_lin, out_const, stream))

"canonicalize!(residual ⋅ residual) + MOI copy on the f64 matrix cores — src/functions.jl:381-386, src/moi_interop.jl:45-62"
quad_gram!(out_quad, out_lin, out_const, A, lda, rows, cols, xvar, b, sign, moi, varmap, workspace, stream) =
    check(ccall((:pmt_quad_gram_f64, lib), Cint,
                (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, Cint, DevPtr, DevPtr, DevPtr, DevPtr, DevPtr, Ptr{Cvoid}),
                A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, out_lin, out_const, workspace, stream))

"bilinearmul!(dest, Q, x', y) — src/functions.jl:840-858"
bilinear!(out_quad, Q, ldq, rows, cols, xvar, yvar, moi, varmap, stream) =
    check(ccall((:pmt_bilinear_f64, lib), Cint, (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, DevPtr, DevPtr, Ptr{Cvoid}),
                Q, ldq, rows, cols, xvar, yvar, moi, varmap, out_quad, stream))

"Matrix{Float64} -> padded device copy (leading dimension ldd >= rows; DESIGN.md §2): a pitched host-to-device copy"
upload_matrix!(plan::Plan, dst::DevPtr, ldd, A::Matrix{Float64}) =
    check(ccall((:pmt_plan_upload_2d, lib), Cint, (Ptr{Cvoid}, DevPtr, Csize_t, Ptr{Cvoid}, Csize_t, Csize_t, Csize_t),
                plan.handle, dst, 8 * ldd, A, 8 * size(A, 1), 8 * size(A, 1), size(A, 2)))

# ---- solver hand-off (DESIGN.md §8): OSQP-style CSC data built on the device, behind MOI.set

"the Gram node with P's CSC values written from the contraction's epilogue (out_quad may be C_NULL)"
quad_gram_csc!(out_P_values, out_quad, out_lin, out_const, A, lda, rows, cols, xvar, b, sign, varmap, alpha, workspace, stream) =
    check(ccall((:pmt_quad_gram_csc_f64, lib), Cint,
                (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, DevPtr, Cdouble, DevPtr, DevPtr, DevPtr, DevPtr, DevPtr, Ptr{Cvoid}),
                A, lda, rows, cols, xvar, b, sign, varmap, alpha, out_P_values, out_quad, out_lin, out_const, workspace, stream))

"structure of a solver matrix from 1-based (row, col) indices (host, once): perm, seg_ptr, colptr, rowval (0-based), nnz"
function csc_order(rows::Vector{Int64}, cols::Vector{Int64}, nrows, ncols; upper::Bool=false)
    n = length(rows)
    perm, seg = zeros(Int64, max(n, 1)), zeros(Int64, n + 1)
    colptr, rowval, nnz = zeros(Int64, ncols + 1), zeros(Int64, max(n, 1)), Ref{Int64}(0)
    check(ccall((:pmt_csc_order, lib), Cint,
                (Int64, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ref{Int64}),
                n, rows, cols, nrows, ncols, upper, perm, seg, colptr, rowval, nnz))
    perm[1:n], seg[1:nnz[] + 1], colptr, rowval[1:nnz[]]
end

"values of a solver matrix: dst[dst_index[s]] = alpha * sum of the MOI coefficients of run s (per re-evaluation)"
csc_values!(dst::DevPtr, src_coeff::DevPtr, stride, nnz_in, perm::DevPtr, seg::DevPtr, nnz_out, alpha, dst_index::DevPtr, stream) =
    check(ccall((:pmt_csc_values_f64, lib), Cint, (DevPtr, Int64, Int64, DevPtr, DevPtr, Int64, Cdouble, DevPtr, DevPtr, Ptr{Cvoid}),
                src_coeff, stride, nnz_in, perm, seg, nnz_out, alpha, dst_index, dst, stream))

"l, u of `f(x) in set` rows from the constants: kind 0 = Zeros/EqualTo, 1 = Nonnegatives/GreaterThan, 2 = Nonpositives/LessThan"
qp_bounds!(l::DevPtr, u::DevPtr, consts::DevPtr, rows, kind, value, infty, stream) =
    check(ccall((:pmt_qp_bounds_f64, lib), Cint, (DevPtr, Int64, Cint, Cdouble, Cdouble, DevPtr, DevPtr, Ptr{Cvoid}),
                consts, rows, kind, value, infty, l, u, stream))

# ---- sparse constraint matrix (SparseMatrixCSC with a fixed pattern; BASELINE config 5)

"row-major order of the structural non-zeros + per-row boundaries of `nslab` column slabs (host, once per pattern)"
function sparse_plan(C::SparseMatrixCSC{Float64,Int64}; nslab::Integer=8)
    m, n = size(C)
    nz = length(C.nzval)
    perm, trow, tcol, rowptr = zeros(Int64, max(nz, 1)), zeros(Int64, max(nz, 1)), zeros(Int64, max(nz, 1)), zeros(Int64, m + 1)
    check(ccall((:pmt_sparse_rowmajor_order, lib), Cint,
                (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
                m, n, C.colptr, C.rowval, perm, trow, tcol, rowptr))
    slab = zeros(Int64, max(m, 1) * (nslab + 1))
    check(ccall((:pmt_sparse_slab_ptr, lib), Cint, (Int64, Int64, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), m, n, nslab, rowptr, tcol, slab))
    (perm = perm, term_col = tcol, row_ptr = rowptr, slab_ptr = slab)
end

"C*x (+|-) d for a sparse C written straight into MOI.VectorAffineTerms (XCD-aware scatter)"
sparse_pack_vector!(out_terms::DevPtr, nzval::DevPtr, perm::DevPtr, term_var::DevPtr, slab_ptr::DevPtr, rows, nslab, varmap::DevPtr,
                    row_offset, stream) =
    check(ccall((:pmt_sparse_pack_vector_slabs_f64, lib), Cint,
                (DevPtr, DevPtr, DevPtr, DevPtr, Int64, Cint, DevPtr, Int64, DevPtr, Ptr{Cvoid}),
                nzval, perm, term_var, slab_ptr, rows, nslab, varmap, row_offset, out_terms, stream))

"the same with UInt32 perm / variable streams (nnz and indices below 2^32: half the index bytes).  Fold the optimizer's index map into
`term_var` (term_var[t] = varmap[x[col]]) and pass `varmap = C_NULL` to drop the per-term map gather; refresh it after mapindices!."
sparse_pack_vector_u32!(out_terms::DevPtr, nzval::DevPtr, perm32::DevPtr, term_var32::DevPtr, slab_ptr::DevPtr, rows, nslab, varmap::DevPtr,
                        row_offset, stream) =
    check(ccall((:pmt_sparse_pack_vector_slabs_u32_f64, lib), Cint,
                (DevPtr, DevPtr, DevPtr, DevPtr, Int64, Cint, DevPtr, Int64, DevPtr, Ptr{Cvoid}),
                nzval, perm32, term_var32, slab_ptr, rows, nslab, varmap, row_offset, out_terms, stream))

"block form of the sparse node (CSC -> row-major through LDS, sparse.hip): band width `cw` (0: use the slab form), column descriptors, the
4-byte index word per term and the per-row band boundaries (host, once per pattern; `plan` from sparse_plan)"
function sparse_block_plan(C::SparseMatrixCSC{Float64,Int64}, plan)
    m, n = size(C)
    nz = length(C.nzval)
    cw = Ref{Cint}(0)
    check(ccall((:pmt_sparse_blocks_width, lib), Cint, (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ref{Cint}), m, n, C.colptr, C.rowval, cw))
    cw[] == 0 && return (cw = 0, desc = UInt64[], idx = UInt32[], band_ptr = Int64[])
    nrb, ncb = cld(m, 128), cld(n, Int(cw[]))
    desc, idx, band = zeros(UInt64, nrb * n), zeros(UInt32, nz), zeros(Int64, m * (ncb + 1))
    check(ccall((:pmt_sparse_blocks_build, lib), Cint,
                (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Cint, Ptr{UInt64}, Ptr{UInt32}, Ptr{Int64}),
                m, n, C.colptr, C.rowval, plan.perm, plan.term_col, plan.row_ptr, cw[], desc, idx, band))
    (cw = Int(cw[]), desc = desc, idx = idx, band_ptr = band)
end

"C*x (+|-) d for a sparse C written straight into MOI.VectorAffineTerms, block form: `col_var[c]` = x[c] (or varmap[x[c]] with `varmap = C_NULL`);
the constants 0 (+|-) d come out of the same launch (`d = C_NULL`, `sign = 0`: zeros; `out_consts = C_NULL`: not written)"
sparse_pack_vector_blocks!(out_terms::DevPtr, out_consts::DevPtr, nzval::DevPtr, desc::DevPtr, idx::DevPtr, band_ptr::DevPtr, col_var::DevPtr, rows, cols,
                           nnz, cw, varmap::DevPtr, row_offset, d::DevPtr, sign, stream) =
    check(ccall((:pmt_sparse_pack_vector_blocks_f64, lib), Cint,
                (DevPtr, DevPtr, DevPtr, DevPtr, DevPtr, Int64, Int64, Int64, Cint, DevPtr, Int64, DevPtr, Cint, DevPtr, DevPtr, Ptr{Cvoid}),
                nzval, desc, idx, band_ptr, col_var, rows, cols, nnz, cw, varmap, row_offset, d, sign, out_terms, out_consts, stream))

"native LinearTerms of the same node, block form"
sparse_assemble_blocks!(out_terms::DevPtr, out_consts::DevPtr, nzval::DevPtr, desc::DevPtr, idx::DevPtr, band_ptr::DevPtr, col_var::DevPtr, rows, cols, nnz,
                        cw, d::DevPtr, sign, stream) =
    check(ccall((:pmt_sparse_assemble_blocks_f64, lib), Cint,
                (DevPtr, DevPtr, DevPtr, DevPtr, DevPtr, Int64, Int64, Int64, Cint, DevPtr, Cint, DevPtr, DevPtr, Ptr{Cvoid}),
                nzval, desc, idx, band_ptr, col_var, rows, cols, nnz, cw, d, sign, out_terms, out_consts, stream))

"out[i] = 0.0 (+|-) d[i] (sign -1 / +1): the constants of C*x (+|-) d (src/functions.jl:751-764)"
consts!(out::DevPtr, d::DevPtr, n, sign, stream) =
    check(ccall((:pmt_consts_f64, lib), Cint, (DevPtr, Int64, Cint, DevPtr, Ptr{Cvoid}), d, n, sign, out, stream))

"dst (cols x rows, leading dimension ldd) = transpose of src (rows x cols, leading dimension lds) — the adjoint rule, src/lazyexpression.jl:206-217"
transpose!(dst::DevPtr, ldd, src::DevPtr, lds, rows, cols, stream) =
    check(ccall((:pmt_transpose_f64, lib), Cint, (DevPtr, Int64, Int64, Int64, DevPtr, Int64, Ptr{Cvoid}), src, lds, rows, cols, dst, ldd, stream))

"(order, groups, stage_rows): the fixed summation order of the node's constant for an r x n problem (include/parametron_hip.h)"
function quad_gram_constant_order(rows, cols)
    o = Ref{Cint}(0); g = Ref{Cint}(0); s = Ref{Cint}(0)
    check(ccall((:pmt_quad_gram_constant_order, lib), Cint, (Int64, Int64, Ref{Cint}, Ref{Cint}, Ref{Cint}), rows, cols, o, g, s))
    Int(o[]), Int(g[]), Int(s[])
end
"bytes of workspace pmt_quad_gram_f64 needs for an r x n problem"
quad_gram_workspace_bytes(rows, cols) = ccall((:pmt_quad_gram_workspace_bytes, lib), Csize_t, (Int64, Int64), rows, cols)

# ---- staged (overlapped) uploads of host-updated Parameters (`Parameter(model, val=buf)`, src/parameter.jl:88): copy stream + commit
stage_upload!(plan::Plan, staging::DevPtr, src::Array) =
    check(ccall((:pmt_plan_stage_upload, lib), Cint, (Ptr{Cvoid}, DevPtr, Ptr{Cvoid}, Csize_t), plan.handle, staging, src, sizeof(src)))
stage_upload_matrix!(plan::Plan, staging::DevPtr, ldd, A::Matrix{Float64}) =
    check(ccall((:pmt_plan_stage_upload_2d, lib), Cint, (Ptr{Cvoid}, DevPtr, Csize_t, Ptr{Cvoid}, Csize_t, Csize_t, Csize_t),
                plan.handle, staging, 8 * ldd, A, 8 * size(A, 1), 8 * size(A, 1), size(A, 2)))
"the same from a FLAT page-locked vector holding the rows x cols value column-major (the staging copy of a Parameter value)"
stage_upload_pitched!(plan::Plan, staging::DevPtr, ldd, src::Vector{Float64}, rows, cols) =
    check(ccall((:pmt_plan_stage_upload_2d, lib), Cint, (Ptr{Cvoid}, DevPtr, Csize_t, Ptr{Cvoid}, Csize_t, Csize_t, Csize_t),
                plan.handle, staging, 8 * ldd, src, 8 * rows, 8 * rows, cols))
commit_staged!(plan::Plan, dst::DevPtr, staging::DevPtr, bytes) =
    check(ccall((:pmt_plan_commit_staged, lib), Cint, (Ptr{Cvoid}, DevPtr, DevPtr, Csize_t), plan.handle, dst, staging, bytes))
staging_consumed!(plan::Plan) = check(ccall((:pmt_plan_staging_consumed, lib), Cint, (Ptr{Cvoid},), plan.handle))
staged_synchronize(plan::Plan) = check(ccall((:pmt_plan_staged_synchronize, lib), Cint, (Ptr{Cvoid},), plan.handle))
"staging slot (0 / 1) of the calls above; alternate it (and the staging buffers) per update so the next copy does not wait for this update's commits"
stage_slot!(plan::Plan, slot::Integer) = check(ccall((:pmt_plan_stage_slot, lib), Cint, (Ptr{Cvoid}, Cint), plan.handle, slot))

"lane of the commits that follow: 0 = the plan's stream, 1 = its side stream (Parameters that only side-lane entries of the tape read)"
commit_lane!(plan::Plan, lane::Integer) = check(ccall((:pmt_plan_commit_lane, lib), Cint, (Ptr{Cvoid}, Cint), plan.handle, lane))

"plan stream: wait for the staged uploads of the current slot (then commit / consume them)"
wait_staged!(plan::Plan) = check(ccall((:pmt_plan_wait_staged, lib), Cint, (Ptr{Cvoid},), plan.handle))

# ---- page-locked host memory: the staging buffers of host-updated Parameters and the arrays a HOST solver is handed
"a Vector{Float64} over page-locked memory (pmt_host_alloc); free with host_free — NOT garbage collected, the device copies into it asynchronously"
function host_alloc(n::Integer)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmt_host_alloc, lib), Cint, (Csize_t, Ref{Ptr{Cvoid}}), 8 * max(n, 1), ref))
    unsafe_wrap(Array, Ptr{Float64}(ref[]), n; own = false)
end
host_free(v::Vector{Float64}) = check(ccall((:pmt_host_free, lib), Cint, (Ptr{Cvoid},), pointer(v)))

"""host_alloc_as(T, n) -> Vector{T} over page-locked memory (pmt_host_alloc) for an isbits T — e.g. the `terms` vector of an MOI function
(`MOI.VectorAffineTerm{Float64}` = 24 bytes = pmt_vector_affine_term): a recorded entry point is given `pointer(v)` as its output and the kernel
stores straight into the vector the function object holds — no device twin, no fetch.  NOT garbage collected (`host_free_ptr(pointer(v))`).
(Registering the host language's own arrays in place — hipHostRegister — was tried in round 6 and removed: small heap arrays share pages
with their neighbours, and unregistering one pulls the mapping from under the others.)"""
function host_alloc_as(::Type{T}, n::Integer) where {T}
    isbitstype(T) || throw(ArgumentError("host_alloc_as: $T is not an isbits type"))
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmt_host_alloc, lib), Cint, (Csize_t, Ref{Ptr{Cvoid}}), sizeof(T) * max(n, 1), ref))
    unsafe_wrap(Array, Ptr{T}(ref[]), n; own = false)
end
host_free_ptr(p::Ptr) = check(ccall((:pmt_host_free, lib), Cint, (Ptr{Cvoid},), p))

# ---- update!(m::Model) of a SMALL model behind one call (include/parametron_hip.h: pmt_model_*; csrc/modelrun.hip)
"the per-solve walk of a small model: mailboxes of host-updated Parameters, seed words of device-regenerated ones, fetches, constants"
mutable struct ModelRun
    handle::Ptr{Cvoid}
    plan::Plan                        # (kept alive: the run refers to it)
    function ModelRun(plan::Plan)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pmt_model_create, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), plan.handle, ref))
        r = new(ref[], plan)
        finalizer(r -> ccall((:pmt_model_destroy, lib), Cint, (Ptr{Cvoid},), r.handle), r)
        r
    end
end
"`host`: the Parameter's value (Matrix: strides (1, size(host, 1)); Vector: cols = 0); `mailbox`: page-locked, `ld` doubles per column.  Returns the slot."
function add_mailbox!(run::ModelRun, host::Ptr{Float64}, rows::Integer, cols::Integer, row_stride::Integer, col_stride::Integer, mailbox::Vector{Float64}, ld::Integer)
    slot = Ref{Cint}(-1)
    check(ccall((:pmt_model_add_mailbox, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Ref{Cint}),
                run.handle, host, rows, cols, row_stride, col_stride, pointer(mailbox), ld, slot))
    Int(slot[])
end
"a device-regenerated Parameter: `word[]` = base + stride * (number of updates so far), stored before every replay"
function add_seed!(run::ModelRun, word::Ref{UInt64}, base::Integer, stride::Integer)
    slot = Ref{Cint}(-1)
    check(ccall((:pmt_model_add_seed, lib), Cint, (Ptr{Cvoid}, Ref{UInt64}, UInt64, UInt64, Ref{Cint}), run.handle, word, base, stride, slot))
    Int(slot[])
end
set_host!(run::ModelRun, slot::Integer, host::Ptr{Float64}) =
    check(ccall((:pmt_model_set_host, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), run.handle, slot, host))
"behind the wait of every update: `dst[] = src[]` (e.g. a scalar function's constant out of its page-locked word)"
add_constant!(run::ModelRun, src::Ptr{Float64}, dst::Ptr{Float64}) =
    check(ccall((:pmt_model_add_constant, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), run.handle, src, dst))
"a result that lives in HBM: copied into `dst` (page-locked) behind every replay, in front of the wait"
add_fetch!(run::ModelRun, dst::Array, src::DevPtr, bytes::Integer = sizeof(dst)) =
    check(ccall((:pmt_model_add_fetch, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Csize_t), run.handle, dst, src, bytes))
num_slots(run::ModelRun) = Int(ccall((:pmt_model_num_slots, lib), Cint, (Ptr{Cvoid},), run.handle))
"""one update!(model): dirty mailboxes / seeds refreshed, the tape replayed, (synchronize) the MOI buffers complete on the host.
`dirty === nothing`: every slot (setdirty!(model) semantics, src/model.jl:132-133); else one byte per slot."""
model_update!(run::ModelRun, dirty::Nothing, synchronize::Bool = true) =
    check(ccall((:pmt_model_update, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), run.handle, C_NULL, 0, synchronize ? 1 : 0))
model_update!(run::ModelRun, dirty::Vector{UInt8}, synchronize::Bool = true) =
    check(ccall((:pmt_model_update, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), run.handle, dirty, length(dirty), synchronize ? 1 : 0))
model_wait!(run::ModelRun) = check(ccall((:pmt_model_wait, lib), Cint, (Ptr{Cvoid},), run.handle))

# ---- delivery to a HOST solver while the re-evaluation runs (include/parametron_hip.h: pmt_plan_record_fetch, pmt_quad_gram_csc_deliver_f64)
"while recording: `dst` (page-locked) receives `bytes` from `src` on the plan's fetch path as soon as what was recorded before it on its lane is done"
record_fetch!(plan::Plan, dst::Array, src::DevPtr, bytes::Integer = sizeof(dst)) =
    check(ccall((:pmt_plan_record_fetch, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Csize_t), plan.handle, dst, src, bytes))
"while recording, pitched: `cols` columns of `rows` doubles out of a padded device matrix (leading dimension `lds`) into host columns `dst_pitch_bytes` apart —
the CSC values of a dense constraint block leave straight out of its Parameter buffer"
record_fetch_matrix!(plan::Plan, dst::Ptr{Float64}, dst_pitch_bytes, src::DevPtr, lds, rows, cols) =
    check(ccall((:pmt_plan_record_fetch_2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, DevPtr, Csize_t, Csize_t, Csize_t),
                plan.handle, dst, dst_pitch_bytes, src, 8 * lds, 8 * rows, cols))
"host -> host: `cols` columns of `rows` doubles, `src_pitch_bytes` apart in the source, `dst_pitch_bytes` apart in the destination, on the library's
worker threads (pmt_host_copy_2d) — the dense blocks of a host solver's A whose Parameter values are already on the host"
host_copy_matrix!(dst::Ptr{Float64}, dst_pitch_bytes, src::Ptr{Float64}, src_pitch_bytes, rows, cols; threads::Integer = 0) =
    check(ccall((:pmt_host_copy_2d, lib), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cvoid}, Csize_t, Csize_t, Csize_t, Cint),
                dst, dst_pitch_bytes, src, src_pitch_bytes, 8 * rows, cols, threads))
"how results leave for the host: 0 automatic, 1 copy engine or an error, 2 kernel copies (include/parametron_hip.h)"
set_host_delivery(mode::Integer) = check(ccall((:pmt_set_host_delivery, lib), Cint, (Cint,), mode))
"(mode, copy engine usable on `device`)"
function host_delivery(device::Integer = 0)
    mode, engine = Ref{Cint}(0), Ref{Cint}(0)
    check(ccall((:pmt_get_host_delivery, lib), Cint, (Cint, Ref{Cint}, Ref{Cint}), device, mode, engine))
    (Int(mode[]), engine[] != 0)
end
"host: every recorded fetch / delivered band group of the last update has landed"
fetch_synchronize(plan::Plan) = check(ccall((:pmt_plan_fetch_synchronize, lib), Cint, (Ptr{Cvoid},), plan.handle))
"pitched device -> host copy on the plan's stream: the columns of a padded device matrix into a dense host column range (setup / serial path)"
fetch_matrix!(plan::Plan, dst::Ptr{Float64}, dst_pitch_bytes, src::DevPtr, lds, rows, cols) =
    check(ccall((:pmt_plan_fetch_2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, DevPtr, Csize_t, Csize_t, Csize_t),
                plan.handle, dst, dst_pitch_bytes, src, 8 * lds, 8 * rows, cols))
"the Gram node with P's CSC values DELIVERED band group by band group into the page-locked `host_P` while the contraction runs"
quad_gram_csc_deliver!(out_P_values, host_P::Vector{Float64}, out_lin, out_const, A, lda, rows, cols, xvar, b, sign, varmap, alpha, workspace, stream; ngroups::Integer = 0) =
    check(ccall((:pmt_quad_gram_csc_deliver_f64, lib), Cint,
                (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, DevPtr, Cdouble, DevPtr, Ptr{Cvoid}, Cint, DevPtr, DevPtr, DevPtr, Ptr{Cvoid}),
                A, lda, rows, cols, xvar, b, sign, varmap, alpha, out_P_values, host_P, ngroups, out_lin, out_const, workspace, stream))

"the Gram node with the MOI quadratic terms DELIVERED row band by row band into page-locked `host_quad` (pmt_host_alloc'ed memory viewed as
24-byte terms) while the contraction runs — the reference's own boundary, src/moi_interop.jl:131-137"
quad_gram_deliver!(out_quad::DevPtr, host_quad::Ptr{Cvoid}, out_lin, out_const, A, lda, rows, cols, xvar, b, sign, moi, varmap, workspace, stream; nstages::Integer = 0) =
    check(ccall((:pmt_quad_gram_deliver_f64, lib), Cint,
                (DevPtr, Int64, Int64, Int64, DevPtr, DevPtr, Cint, Cint, DevPtr, DevPtr, Ptr{Cvoid}, Cint, DevPtr, DevPtr, DevPtr, Ptr{Cvoid}),
                A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, host_quad, nstages, out_lin, out_const, workspace, stream))

# ---- further builders used by ParametronHIPBackend.jl
"x (+|-) v for x::Vector{Variable} (bounds): one term per row; native and/or MOI output may be C_NULL — src/functions.jl:421,751-764"
vars_addsub!(out_lt::DevPtr, out_vat::DevPtr, out_consts::DevPtr, xvar::DevPtr, n, v::DevPtr, sign, varmap::DevPtr, row_offset, stream) =
    check(ccall((:pmt_vars_addsub_f64, lib), Cint, (DevPtr, Int64, DevPtr, Cint, DevPtr, Int64, DevPtr, DevPtr, DevPtr, Ptr{Cvoid}),
                xvar, n, v, sign, varmap, row_offset, out_lt, out_vat, out_consts, stream))
"dest[i] = s * y[i] for a device scalar s: scale!(dest, x::Number, y::Vector{AffineFunction}) — src/functions.jl:895-915"
affvec_scale!(out_terms::DevPtr, out_consts::DevPtr, rows, nterms, y_terms::DevPtr, y_consts::DevPtr, s_dev::DevPtr, stream) =
    check(ccall((:pmt_affvec_scale_f64, lib), Cint, (Int64, Int64, DevPtr, DevPtr, DevPtr, Cdouble, DevPtr, DevPtr, Ptr{Cvoid}),
                rows, nterms, y_terms, y_consts, s_dev, 0.0, out_terms, out_consts, stream))
"update!(::MOI.VectorAffineFunction, fs, varmap) of a materialised Vector{AffineFunction} with uniform rows — src/moi_interop.jl:64-81"
pack_vector_affine!(out_terms::DevPtr, terms::DevPtr, rows, row_len, varmap::DevPtr, row_offset, stream) =
    check(ccall((:pmt_pack_vector_affine_f64, lib), Cint, (DevPtr, DevPtr, Int64, Int64, DevPtr, Int64, DevPtr, Ptr{Cvoid}),
                terms, C_NULL, rows, row_len, varmap, row_offset, out_terms, stream))
"device-to-device copy on the stream (vcat! pieces of constants)"
copy_bytes!(dst::DevPtr, src::DevPtr, bytes, stream) =
    check(ccall((:pmt_copy_bytes, lib), Cint, (DevPtr, DevPtr, Csize_t, Ptr{Cvoid}), dst, src, bytes, stream))

# ---- batched independent models across GPUs (BASELINE config 4): the library's own RCCL communicator
"rank 0: a fresh 128-byte id; carry it to the other ranks with whatever launcher is at hand (MPI.jl, Distributed, a file)"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:pmt_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    id
end
function comm_init_rank(nranks::Integer, rank::Integer, id::Vector{UInt8}, device::Integer)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmt_comm_init_rank, lib), Cint, (Cint, Cint, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), nranks, rank, id, device, ref))
    ref[]
end
comm_destroy(comm::Ptr{Cvoid}) = check(ccall((:pmt_comm_destroy, lib), Cint, (Ptr{Cvoid},), comm))
"(per_rank, first) of `rank`'s shard of `total` instances; DimensionMismatch when the batch does not divide over the ranks"
function batch_shard(total::Integer, nranks::Integer, rank::Integer)
    per = Ref{Int64}(0); first = Ref{Int64}(0)
    check(ccall((:pmt_batch_shard, lib), Cint, (Int64, Cint, Cint, Ref{Int64}, Ref{Int64}), total, nranks, rank, per, first))
    per[], first[]
end
"one re-evaluation of this rank's instances, chunk c on the wire while chunk c + 1 is computed; `gathered` ends up with every rank's slabs"
batch_step!(comm::Ptr{Cvoid}, A::DevPtr, b::DevPtr, C::DevPtr, d::DevPtr, per_rank, n, r, m, local_slabs::DevPtr, gathered::DevPtr, stride, chunk, stream) =
    check(ccall((:pmt_batch_step_f64, lib), Cint,
                (Ptr{Cvoid}, DevPtr, DevPtr, DevPtr, DevPtr, Int64, Int64, Int64, Int64, Cint, Cint, DevPtr, DevPtr, Int64, Int64, Ptr{Cvoid}),
                comm, A, b, C, d, per_rank, n, r, m, -1, -1, local_slabs, gathered, stride, chunk, stream))

"the same callback RECORDED with its seed in a host word (`Ref{UInt64}`, kept alive by the caller): every replay draws the stream of the
seed the word holds then — `seed[] += 1000` before `update!(plan)` is the whole callback"
device_uniform_dyn!(dst::DevPtr, rows, cols, lda, seed::Ref{UInt64}, scale, stream) =
    check(ccall((:pmt_fill_uniform_dyn_f64, lib), Cint, (DevPtr, Int64, Int64, Int64, Ref{UInt64}, Cdouble, Ptr{Cvoid}), dst, rows, cols, lda, seed, scale, stream))

"device-side `rand!` Parameter callback (README.md:36-43): U[0,1)*scale, counter based"
device_uniform!(dst::DevPtr, n, seed, scale, stream) =
    check(ccall((:pmt_fill_uniform_f64, lib), Cint, (DevPtr, Int64, UInt64, Cdouble, Ptr{Cvoid}), dst, n, seed, scale, stream))

end # module
