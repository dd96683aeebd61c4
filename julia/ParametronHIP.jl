# ParametronHIP.jl — the reference-side binding a Parametron.jl maintainer would add to route the update! hot path
# through libparametron_hip.so (include/parametron_hip.h).  NOT EXECUTED in this repository's CI: the build image has no
# Julia (SURVEY.md §0.4); the same C ABI is exercised from Python (parametron.jl_amd/_lib.py) by tests/.
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
recording_stream(plan::Plan) = ccall((:pmt_plan_recording_stream, lib), Ptr{Cvoid}, (Ptr{Cvoid},), plan.handle)
begin_record!(plan::Plan) = check(ccall((:pmt_plan_begin_record, lib), Cint, (Ptr{Cvoid},), plan.handle))
end_record!(plan::Plan) = check(ccall((:pmt_plan_end_record, lib), Cint, (Ptr{Cvoid},), plan.handle))
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
bilinear!(out_quad, Q, rows, cols, xvar, yvar, moi, varmap, stream) =
    check(ccall((:pmt_bilinear_f64, lib), Cint, (DevPtr, Int64, Int64, DevPtr, DevPtr, Cint, DevPtr, DevPtr, Ptr{Cvoid}),
                Q, rows, cols, xvar, yvar, moi, varmap, out_quad, stream))

"device-side `rand!` Parameter callback (README.md:36-43): U[0,1)*scale, counter based"
device_uniform!(dst::DevPtr, n, seed, scale, stream) =
    check(ccall((:pmt_fill_uniform_f64, lib), Cint, (DevPtr, Int64, UInt64, Cdouble, Ptr{Cvoid}), dst, n, seed, scale, stream))

end # module
