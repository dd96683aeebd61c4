/*
 * parametron_hip.h — C ABI of libparametron_hip.so: the MI355X (gfx950) implementation of
 * Parametron.jl's parameter-update / coefficient re-evaluation hot path.
 *
 * The reference (tkoolen/Parametron.jl v0.9.1, pure Julia) has no FFI of its own; the de-facto
 * operator interface of the path is
 *   (1) the in-place builders of Parametron.Functions that the `optimize` rewrite rules splice
 *       into the lazy-expression DAG           (src/lazyexpression.jl:200-302, src/functions.jl)
 *   (2) update!(moi_f, f, varmap) x3            (src/moi_interop.jl:35-81)
 *   (3) the per-node call ABI FunctionWrapper{T,Tuple{}} (src/FunctionWrappersQuickFix.jl:108-126)
 *       driven by update!(m::Model)             (src/model.jl:132-143).
 * Each entry point below names the reference function(s) it replaces.  A Julia host binds them
 * with `ccall((:pmt_xxx, libparametron_hip), Cint, (...), ...)` — see INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no HIP/torch types: `void *stream` is a hipStream_t (NULL = default stream);
 *     every `const double *`, term pointer etc. is a DEVICE pointer unless named host_*.
 *   - all calls are asynchronous on `stream`; nothing allocates or synchronises unless stated
 *     (the reference's `@allocated solve!(model) == 0` contract, test/model.jl:116-124).
 *   - matrices are column-major with leading dimension `lda` (Julia Matrix{Float64},
 *     src/functions.jl:790-796); variable indices are 1-based Int64 (Variable.index).
 *   - `varmap` is model_var_to_optimizer (src/model.jl:8,100-107): varmap[k-1] is the optimizer
 *     index of Variable k; NULL = IdentityVarMap (src/moi_interop.jl:32-33).
 *   - return value: PMT_OK or an error code; pmt_last_error() gives the message.  The Julia/Python
 *     host maps PMT_DIMENSION_MISMATCH -> DimensionMismatch, PMT_INVALID_ARGUMENT -> ArgumentError,
 *     PMT_STATE_ERROR -> ErrorException (src/functions.jl:780-781, src/model.jl:50,61,69).
 *   - threading: calls on one stream/plan must be externally serialised (the reference is
 *     single-threaded and shares `dest` buffers, src/lazyexpression.jl:202-203).
 */
#ifndef PARAMETRON_HIP_H
#define PARAMETRON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMT_OK 0
#define PMT_DIMENSION_MISMATCH 1
#define PMT_INVALID_ARGUMENT 2
#define PMT_HIP_ERROR 3
#define PMT_STATE_ERROR 4
#define PMT_OUT_OF_MEMORY 5

/* Term layouts = the Julia isbits structs. */
typedef struct { double coeff; int64_t var; } pmt_linear_term;                 /* LinearTerm{Float64} src/functions.jl:110-113 == MOI.ScalarAffineTerm (moi_interop.jl:40) */
typedef struct { double coeff; int64_t row; int64_t col; } pmt_quadratic_term; /* QuadraticTerm{Float64} src/functions.jl:136-140 == MOI.ScalarQuadraticTerm (moi_interop.jl:59) */
typedef struct { int64_t output_index; double coeff; int64_t var; } pmt_vector_affine_term; /* MOI.VectorAffineTerm (moi_interop.jl:75) */

const char *pmt_last_error(void);
int pmt_version(void);
/* number of visible HIP devices (<= 0: none); used by hosts to fail loudly without a GPU */
int pmt_device_count(void);

/* ---------------------------------------------------------------------------------------
 * Dense affine nodes
 * ------------------------------------------------------------------------------------- */

/* y = A*x (+|-) b as Vector{AffineFunction}: out_terms[row*cols + col] = (A[row,col], xvar[col]),
 * out_consts[row] = 0.0 (+|-) b[row].   sign: +1 vecadd!, -1 vecsubtract!, 0 no vector (b ignored).
 * Replaces matvecmul!(y, A, x::Vector{Variable}) src/functions.jl:775-798 fused with
 * vecadd!/vecsubtract!(dest, y, b) :751-764 (copyto! :422-427, add!/subtract! Number :452,:474). */
int pmt_affine_assemble_f64(const double *A, int64_t lda, int64_t rows, int64_t cols,
                            const int64_t *xvar, const double *b, int sign,
                            pmt_linear_term *out_terms, double *out_consts, void *stream);

/* Same node written straight into MOI.VectorAffineFunction buffers:
 * out_terms[row*cols + col] = (row_offset + row + 1, A[row,col], varmap[xvar[col]]), out_consts as above.
 * Replaces the chain above + update!(::MOI.VectorAffineFunction, fs, varmap) src/moi_interop.jl:64-81. */
int pmt_affine_pack_vector_f64(const double *A, int64_t lda, int64_t rows, int64_t cols,
                               const int64_t *xvar, const double *b, int sign,
                               const int64_t *varmap, int64_t row_offset,
                               pmt_vector_affine_term *out_terms, double *out_consts, void *stream);
/* The same node in its BACKGROUND form (identical output): a kernel of at most 16 VGPRs and no LDS, slow on an idle chip but co-resident
 * with the persistent contraction of pmt_quad_gram_f64 (which leaves 16 of a SIMD's 512 VGPRs free).  Recorded on a plan's side lane
 * (pmt_plan_set_lane) it runs inside the contraction instead of behind it. */
int pmt_affine_pack_vector_background_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b,
                                          int sign, const int64_t *varmap, int64_t row_offset, pmt_vector_affine_term *out_terms,
                                          double *out_consts, void *stream);

/* dest[i] = x[i] (+|-) v[i] for x::Vector{Variable}: one term (1.0, xvar[i]) per row, constant 0.0 (+|-) v[i]
 * (`x - l` bounds, test/model.jl:162-163).  vecadd!/vecsubtract! with copyto!(f, ::Variable)
 * src/functions.jl:421.  Native output (out_terms_lt) and/or MOI output (out_terms_vat) may be NULL. */
int pmt_vars_addsub_f64(const int64_t *xvar, int64_t n, const double *v, int sign,
                        const int64_t *varmap, int64_t row_offset,
                        pmt_linear_term *out_terms_lt, pmt_vector_affine_term *out_terms_vat,
                        double *out_consts, void *stream);

/* ---------------------------------------------------------------------------------------
 * Generic term-list nodes (materialised Vector{AffineFunction} = flat LT buffer + row_ptr + consts).
 * row_ptr has rows+1 entries (device); NULL means uniform rows of `row_len` terms.
 * ------------------------------------------------------------------------------------- */

/* dst row i = [ xa[i].linear ; sb * xb[i].linear ],  constant  ca[i] (+ sb*cb[i])  with sb = +1 or -1 (exact):
 *   copyto! (:422-427)  then  add!(f, ::AffineFunction) :455  /  subtract!(f, ::AffineFunction) :477-485,
 * i.e. vecadd!/vecsubtract! :751-764 on any mix of Vector{AffineFunction}, Vector{Variable} (pre-materialised
 * by the host as one (1.0, var) term per row, consts NULL: copyto!(f, ::Variable) :421 leaves the constant 0)
 * and number vectors (terms NULL, consts = the numbers: copyto!(f, ::Number) :419, add!/subtract! :452,:474).
 * A part with consts == NULL contributes no constant; part b may be absent altogether (plain copyto!, vcat!
 * pieces :969-994).  If part a has no constants the row constant is 0.0 (+ sb*cb[i]). */
int pmt_affvec_combine_f64(int64_t rows,
                           const pmt_linear_term *xa_terms, const int64_t *xa_row_ptr, int64_t xa_row_len, const double *xa_consts,
                           const pmt_linear_term *xb_terms, const int64_t *xb_row_ptr, int64_t xb_row_len, const double *xb_consts, int sb,
                           pmt_linear_term *out_terms, const int64_t *out_row_ptr, int64_t out_row_len, double *out_consts,
                           void *stream);

/* dest[i] = s * y[i] where s is a device scalar (Parameter{Float64}) or host constant:
 * scale!(dest, x::Number, y::Vector{AffineFunction}) src/functions.jl:895-915 -> mul! :578 -> muladd! :515-523.
 * coeff = s*coeff, const = 0 + y.c*s.  s_dev == NULL uses s_host. */
int pmt_affvec_scale_f64(int64_t rows, int64_t nterms, const pmt_linear_term *y_terms, const double *y_consts,
                         const double *s_dev, double s_host,
                         pmt_linear_term *out_terms, double *out_consts, void *stream);

/* dest[i] = (s, yvar[i]): scale!(dest::Vector{LinearTerm}, x::Number, y::Vector{Variable}) src/functions.jl:873-893 */
int pmt_scale_vars_f64(const int64_t *yvar, int64_t n, const double *s_dev, double s_host, pmt_linear_term *out_terms, void *stream);
/* dest .= s .* y for number arrays: scale! src/functions.jl:917-925 */
int pmt_scale_numbers_f64(const double *y, int64_t n, const double *s_dev, double s_host, double *out, void *stream);
/* dest[j, i] = A[i, j] — the closure of the `adjoint` rewrite rule src/lazyexpression.jl:206-217.
 * src is rows x cols (leading dimension lds), dst is cols x rows (leading dimension ldd), both column-major. */
int pmt_transpose_f64(const double *src, int64_t lds, int64_t rows, int64_t cols, double *dst, int64_t ldd, void *stream);
/* quadratic term lists: out = [ qa ; sb*qb ], sb = +1/-1: copyto! src/functions.jl:434-439, add! :459, subtract! :492-500 */
int pmt_quad_combine_f64(const pmt_quadratic_term *qa, int64_t na, const pmt_quadratic_term *qb, int64_t nb, int sb,
                         pmt_quadratic_term *out, void *stream);
/* out[i] = (s * q[i].coeff, row, col): muladd!(dest::QuadraticFunction, x::QuadraticFunction, y::Number) :526-534 */
int pmt_quad_scale_f64(const pmt_quadratic_term *q, int64_t n, const double *s_dev, double s_host, pmt_quadratic_term *out, void *stream);
/* device-to-device copy on the stream (array-of-references copyto! nodes, src/lazyexpression.jl:280-282) */
int pmt_copy_bytes(void *dst, const void *src, size_t bytes, void *stream);

/* y = A * X with X::Vector{AffineFunction} of uniform length L:
 * row `row` = concat over col of A[row,col]*X[col].linear ; const = sum_col X.c[col]*A[row,col] (in col order).
 * matvecmul!(y, A, x::Vector{AffineFunction}) src/functions.jl:800-822 (muladd! :524 -> :515-523). */
int pmt_matvecmul_affs_f64(const double *A, int64_t lda, int64_t rows, int64_t cols,
                           const pmt_linear_term *x_terms, int64_t x_row_len, const double *x_consts,
                           pmt_linear_term *out_terms, double *out_consts, void *stream);

/* dot(v, x) family producing an AffineFunction (src/functions.jl:665-687):
 *   Number[] . Variable[]        -> out_terms[i] = (v[i], xvar[i]), const 0            (:676-687)
 *   Number[] . AffineFunction[]  -> concat_i v[i]*X[i].linear, const = sum_i X.c[i]*v[i] (:665-674, uniform L) */
int pmt_vecdot_numbers_vars_f64(const double *v, const int64_t *xvar, int64_t n,
                                pmt_linear_term *out_terms, double *out_const, void *stream);
int pmt_vecdot_numbers_affs_f64(const double *v, int64_t n,
                                const pmt_linear_term *x_terms, int64_t x_row_len, const double *x_consts,
                                pmt_linear_term *out_terms, double *out_const, void *stream);

/* ---------------------------------------------------------------------------------------
 * Quadratic nodes
 * ------------------------------------------------------------------------------------- */

/* dest = x . y for x, y ::Vector{AffineFunction} with uniform row lengths nx, ny (LITERAL expansion):
 *   quad[(i*nx + a)*ny + b] = (x[i].lin[a].coeff * y[i].lin[b].coeff, x[i].lin[a].var, y[i].lin[b].var)
 *   lin[i*(nx+ny) + a]      = (y.c[i] * x[i].lin[a].coeff, var)        a < nx
 *   lin[i*(nx+ny) + nx + b] = (x.c[i] * y[i].lin[b].coeff, var)        b < ny
 *   const = sum_i x.c[i]*y.c[i]  accumulated left to right
 * = _vecdot!(dest::QuadraticFunction, x, y) src/functions.jl:702-709 over muladd! :548-576.
 * moi != 0 fuses update!(::MOI.ScalarQuadraticFunction, f, varmap) src/moi_interop.jl:45-62:
 * indices go through varmap and coefficients with rowvar == colvar are doubled (:58). */
int pmt_quad_expand_f64(int64_t rows,
                        const pmt_linear_term *x_terms, int64_t nx, const double *x_consts,
                        const pmt_linear_term *y_terms, int64_t ny, const double *y_consts,
                        int moi, const int64_t *varmap,
                        pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                        void *stream);

/* Canonical least-squares objective for residual = A*x (+|-) b:  canonicalize!(residual . residual)
 * (src/functions.jl:381-386 applied to the literal result above), then the MOI copy:
 *   out_quad[tri(j,k)] = (2 * sum_i A[i,j]*A[i,k], vm[xvar[j]], vm[xvar[k]])  for j <= k, row-major upper triangle
 *   out_lin[j]         = (2 * sum_i c_i*A[i,j], vm[xvar[j]])   with c_i = 0.0 (+|-) b[i]
 *   out_const          = sum_i c_i^2 in a FIXED order that (rows, cols) alone determine (pmt_quad_gram_constant_order below): the
 *                        reference's left-to-right sum (src/functions.jl:574, bit for bit), 2048 interleaved chains, or the fused
 *                        forms' per-workgroup order — every one within (rows / 2048 + 2048) * eps / 2 of the exact sum.
 * Shapes of up to 2048 columns take the fused forms of csrc/gram_tall.hip whatever the row count (the triangle of every diagonal
 * 128-column tile, out_lin and out_const from ONE pass over A; the strictly upper tiles from one stream-K launch); tiny shapes recorded
 * into a plan are nodes of its one-launch interpreter (csrc/small.hip); 2049 .. 4096 columns below 2^29 elements (config 2) the one-launch
 * form of csrc/gram_mid.hip (round 6c; the staged `_deliver_` entry points keep the stream-K kernel there, with out_lin within 1e-13 of the
 * plain call's and out_const in the same order); wider or larger shapes the stream-K node with the two reductions on a side stream.
 * A is read as lda x cols doubles: a kernel may read (and ignore) the padding rows rows .. lda - 1 of a column, the last one included.
 * Requires xvar strictly increasing (distinct variables in sorted order — what Variable(model) yields);
 * moi == 0 keeps native indices (no varmap) but the same coefficients as the MOI form are NOT produced:
 *   native canonical form has diagonal coefficient (A'A)[j,j] and off-diagonal 2*(A'A)[j,k].
 * f64 MFMA contraction; `workspace` (device, pmt_quad_gram_workspace_bytes) holds split-K partial tiles and the chunk / chain sums. */
size_t pmt_quad_gram_workspace_bytes(int64_t rows, int64_t cols);
/* The summation order of the node's constant c'c for an r x n problem — fixed by (rows, cols) alone, reported so that a caller (and the
 * parity tests) can restate it:
 *   order 0  sequential, the reference's left-to-right sum (src/functions.jl:574), bit for bit: tiny shapes, and the stream-K node (beyond
 *            4096 columns or from 2^29 elements) where the contraction hides the one-wave chain
 *   order 1  `groups` = 2048 interleaved chains (chain t adds rows t, t + 2048, .. in order), chain totals added left to right: the stream-K
 *            node with rows > 8192, or a sequential chain that would take half as long as the contraction beside it or longer
 *   order 2  the fused forms (cols <= 2048; csrc/gram_tall.hip): `groups` workgroups, workgroup g takes the stages g, g + groups, .. of
 *            `stage_rows` rows; per stage eight row-pair lanes (rows 16 j + 2 p, + 1) add their squares in row order, an 8-lane tree
 *            ((0+4)+(2+6))+((1+5)+(3+7)) closes a workgroup, the workgroups are added in 16 interleaved slices, then the slices
 *   order 3  the same with SIXTEEN row-pair lanes (rows 32 j + 2 p, + 1; a 16-lane tree): the 16-column panel, cols <= 16
 *            (cols <= 64: below 32768 rows)
 *   order 4  narrow panels, cols <= 64, from 32768 rows (gram_stream_kernel): iterations of `stage_rows` rows dealt out to the 4 * `groups` WAVES (wave
 *            4 g + w: iterations 4 g + w, + 4 * groups, ..); contraction slot k of a wave adds rows 8 i + 2 k, + 1 of its iterations in order;
 *            slots (0 + 2) + (1 + 3); the four waves of a workgroup in order; workgroups in 16 interleaved slices, then the slices
 *   order 5  wide shapes in one launch (gram_mid.hip: 129 .. 4096 columns — from 384 columns everything below 2^29 elements, below 384 / 320 / 193
 *            columns up to 2^27 / 2^26 / 2^25; gram.hip: gram_mid_applies — config 2 since round 6c): `stage_rows` = 512 strided chains (thread
 *            t adds rows t, t + 512, ..), a shuffle tree (32, 16, .., 1) per 64 threads, the eight results in order
 * (tests/gpu_util.py restates every order bit for bit.)
 * Every order is within (rows / 2048 + 2048) * eps / 2 relative of the exact sum for same-signed terms: far inside the 1e-12 parity bar. */
int pmt_quad_gram_constant_order(int64_t rows, int64_t cols, int *order, int *groups, int *stage_rows);
int pmt_quad_gram_f64(const double *A, int64_t lda, int64_t rows, int64_t cols,
                      const int64_t *xvar, const double *b, int sign,
                      int moi, const int64_t *varmap,
                      pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                      void *workspace, void *stream);
/* The same node with the solver hand-off fused into the epilogue (see "Solver hand-off" below): the values of P's upper
 * triangle in CSC order, out_P_values[k(k+1)/2 + j] = alpha * 2 * sum_i A[i,j]*A[i,k] (j <= k) — valid as the CSC value array
 * whenever j -> vm[xvar[j]] is strictly increasing (col_ptr[c] counts the variables below, row indices follow).  out_quad
 * (MOI terms, as above with moi = 1) is optional here: NULL skips the 24-byte term structs entirely.  out_lin / out_const as above.
 * The values equal the coefficients pmt_quad_gram_f64 writes for the same inputs bit for bit, except for TINY shapes (those the small-plan
 * interpreter takes: row-order sums there, the contraction's MFMA order here — equal to rounding; the constant is sequential in both). */
int pmt_quad_gram_csc_f64(const double *A, int64_t lda, int64_t rows, int64_t cols,
                          const int64_t *xvar, const double *b, int sign, const int64_t *varmap,
                          double alpha, double *out_P_values, pmt_quadratic_term *out_quad,
                          pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream);

/* The same node (values only, no term structs) with the values additionally DELIVERED TO THE HOST while the contraction runs — the
 * reference's boundary is a host solver (MOI.set, src/moi_interop.jl:131-137; OSQP's update_P takes exactly this array).  host_P_values
 * is page-locked host memory (pmt_host_alloc) of cols*(cols+1)/2 doubles.  The contraction runs in STAGES over the column bands of P
 * (`ngroups` stages, 0 = default: half a grid's worth of 128 x 128 tiles per stage, at most 16), walked from the LAST band to the first (the
 * long bands leave while the contraction is still busy, the short ones are what is left at its end): each stage is a launch over all CUs
 * with its tiles split in two along the contraction — the two workgroups of a tile exchange halves through the workspace and each finishes
 * one row half (other stage sizes: a second launch adds the partial sums) — and the copy engine (HSA SDMA; a courier kernel where that is
 * not available) ships the bands a stage has completed while the next stage is computed, so P leaves at PCIe speed from the first stage on.
 * Splitting a tile changes its summation order (two half sums added): out_P_values here and from pmt_quad_gram_csc_f64 agree to rounding
 * (a few ulp), not bit for bit; both are deterministic, and host_P_values is out_P_values of the same call bit for bit.  `stream` does not
 * wait for the delivery: pmt_plan_fetch_synchronize (or pmt_fetch_synchronize for a plain stream) does — PMT_HIP_ERROR there if a transfer
 * never started (10 s; courier: no progress for 2 s); the next call on the same stream waits by itself until the previous delivery has
 * read out_P_values. */
int pmt_quad_gram_csc_deliver_f64(const double *A, int64_t lda, int64_t rows, int64_t cols,
                                  const int64_t *xvar, const double *b, int sign, const int64_t *varmap,
                                  double alpha, double *out_P_values, double *host_P_values, int ngroups,
                                  pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream);
/* pmt_quad_gram_f64 with the QUADRATIC TERMS delivered to the host the same way — the reference's own boundary (MOI.set of the objective's
 * ScalarQuadraticFunction, src/moi_interop.jl:131-137): host_quad is page-locked memory for cols*(cols+1)/2 pmt_quadratic_term; the stages run
 * over tile ROW bands (the term array is row-major), each completed band range leaves while the next stage is computed.  Same remarks as above
 * (stages, summation order of split tiles, pmt_plan_fetch_synchronize / pmt_fetch_synchronize). */
int pmt_quad_gram_deliver_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                              int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, pmt_quadratic_term *host_quad, int nstages,
                              pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream);
/* host: block until every copy on the fetch stream of `stream` (a HIP stream, not a recording handle) has landed.  PMT_HIP_ERROR when a
 * delivery failed on the device: a transfer that never started, a courier without progress, or a split tile of a staged contraction whose
 * first half never arrived (the tile is then NaN in out_P_values / out_quad — never a plausible half sum — and this call says so). */
int pmt_fetch_synchronize(void *stream);

/* How results leave for the host (recorded fetches, the deliveries above), process-wide, from the next replay / call on:
 *   0  automatic (default): the copy engine (HSA SDMA transfers started by signals the kernels set) when the process's HSA runtime and the
 *      device's agent are found — matched by PCI domain:bus:device, an ambiguous match counts as none — kernel copies otherwise;
 *   1  the copy engine or PMT_STATE_ERROR (a deployment that must not fall back silently);
 *   2  kernel copies (a courier kernel for deliveries, copy kernels for recorded fetches) on the stream's fetch stream.
 * Same bytes in the host arrays either way.  pmt_get_host_delivery reports the mode and whether `device` has a usable copy engine. */
int pmt_set_host_delivery(int mode);
int pmt_get_host_delivery(int device, int *out_mode, int *out_copy_engine);
/* TEST HOOK (fault injection; 0 = off, the default).  1: in a staged contraction the first halves of split tiles never announce
 * themselves, and the second halves' bounded wait is cut from 2 s to 20 ms — exercises the error path of pmt_fetch_synchronize above.
 * 2: the grid barriers of a small plan's run on several workgroups wait (20 ms) for an arrival that never comes: pmt_plan_synchronize
 * returns PMT_HIP_ERROR for that re-evaluation.  4: every run on several workgroups is launched on ONE (the path a plan takes while another
 * plan's multi-workgroup run is in flight on the device).  Bits combine. */
int pmt_set_fault_injection(int what);

/* dest = transpose(x) * Q * y:  quad[k] = (Q[k] (column-major linear index), x[k / ny], y[k % ny])
 * bilinearmul! src/functions.jl:840-858 (the Q' pairing quirk is reproduced).
 * The linear index k is that of the rows x cols matrix; ldq is the leading dimension of its device copy.  moi as above. */
int pmt_bilinear_f64(const double *Q, int64_t ldq, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *yvar,
                     int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, void *stream);

/* x . y for Variable/LinearTerm vectors: quad[i] = (xc[i]*yc[i], xvar[i], yvar[i]); xc/yc NULL = Variables (coeff 1)
 * _vecdot! src/functions.jl:689-700. */
int pmt_vecdot_terms_f64(int64_t n, const double *xc, const int64_t *xvar, const double *yc, const int64_t *yvar,
                         int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, void *stream);

/* AffineFunction[] . Variable[] (uniform row length L):
 *   quad[i*L + a] = (x[i].lin[a].coeff, x[i].lin[a].var, yvar[i]) ; lin[i] = (x.c[i], yvar[i])
 * _vecdot! :702-709 over muladd!(dest, ::AffineFunction, ::Variable) src/functions.jl:537-546. */
int pmt_vecdot_affs_vars_f64(int64_t rows, const pmt_linear_term *x_terms, int64_t L, const double *x_consts,
                             const int64_t *yvar, int moi, const int64_t *varmap,
                             pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, void *stream);

/* ---------------------------------------------------------------------------------------
 * MOI copies of materialised native functions — update!(moi_f, f, varmap) src/moi_interop.jl:35-81
 * ------------------------------------------------------------------------------------- */
int pmt_pack_scalar_affine_f64(const pmt_linear_term *terms, int64_t n, const int64_t *varmap,
                               pmt_linear_term *out_terms, void *stream);                       /* :35-43 */
int pmt_pack_scalar_quadratic_f64(const pmt_quadratic_term *quad, int64_t nq, const int64_t *varmap,
                                  pmt_quadratic_term *out_quad, void *stream);                  /* :53-60 (diag x2) */
int pmt_pack_vector_affine_f64(const pmt_linear_term *terms, const int64_t *row_ptr, int64_t rows, int64_t row_len,
                               const int64_t *varmap, int64_t row_offset,
                               pmt_vector_affine_term *out_terms, void *stream);                /* :64-81 */

/* ---------------------------------------------------------------------------------------
 * Generic canonicalize! (src/functions.jl:269-272, 381-386; sort_and_combine! src/util.jl:9-26) for arbitrary term lists.
 * Indices are static on this path, so the sort happens once on the host (pmt_canonical_order_*: permutation by canonical key,
 * run boundaries seg_ptr[nseg+1], and the indices of the combined terms — a run of one keeps its original (row, col), util.jl:18-19);
 * per re-evaluation pmt_segment_sum_f64 adds the coefficients of each run (duplicates in original order; the reference's order is
 * that of its unstable QuickSort, so coefficients agree to rounding) and writes ONLY the coefficient field (offset 0) of each
 * output term, whose index fields the host wrote when the node was created.
 * ------------------------------------------------------------------------------------- */
int pmt_canonical_order_affine(int64_t n, const int64_t *host_vars, int64_t *host_perm, int64_t *host_seg_ptr, int64_t *host_out_vars,
                               int64_t *nseg);
int pmt_canonical_order_quadratic(int64_t n, const int64_t *host_rows, const int64_t *host_cols, int64_t *host_perm, int64_t *host_seg_ptr,
                                  int64_t *host_out_rows, int64_t *host_out_cols, int64_t *nseg);
int pmt_segment_sum_f64(const void *in_terms, int64_t in_stride_bytes, const int64_t *perm, const int64_t *seg_ptr, int64_t nseg,
                        void *out_terms, int64_t out_stride_bytes, void *stream);
/* The same ordering computed ON THE DEVICE from the term buffer where it lies (no copy of the indices to the host, no single-threaded
 * sort): a stable radix sort by the packed canonical key, the run boundaries by stream compaction.  perm: int64[n], seg_ptr: int64[n + 1],
 * both device; *nseg_host receives the number of distinct terms (the only bytes that cross PCIe — the host sizes the output from them).
 * Setup-time call on a HIP stream (synchronises).  PMT_INVALID_ARGUMENT when an index does not fit the packed key (>= 2^32): use the
 * host functions above.  pmt_canonical_init_terms then writes the static part of the canonical function, out_terms[s] = (0.0, indices of
 * run s), with the reference's conventions (a run of one keeps its original (row, col), util.jl:18-19). */
int pmt_canonical_order_device(const void *terms, int64_t n, int term_bytes, int64_t *perm, int64_t *seg_ptr, int64_t *nseg_host, void *stream);
int pmt_canonical_init_terms(const void *terms, int term_bytes, const int64_t *perm, const int64_t *seg_ptr, int64_t nseg, void *out_terms, void *stream);

/* prune_zero!(f; atol) (src/functions.jl:294-297, 409-413): out_terms = the terms with abs(coeff) > atol, in order; *out_count (DEVICE
 * memory) = how many.  term_bytes: 16 (pmt_linear_term) or 24 (pmt_quadratic_term).  Not on the solve path: the count is data
 * dependent, the caller synchronises before using it.  NOTE the reference's quirk: prune_zero!(::QuadraticFunction; atol) prunes the
 * affine part with the DEFAULT atol (0), :410 — callers that mirror it pass atol only for the quadratic terms. */
size_t pmt_prune_zero_workspace_bytes(int64_t n, int term_bytes);
int pmt_prune_zero_f64(const void *terms, int64_t n, int term_bytes, double atol, void *out_terms, int64_t *out_count,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Solver hand-off: what happens AFTER MOI.set(optimizer, ...) (src/moi_interop.jl:134,171) — in the
 * reference third-party code (MathOptInterface 0.8 + the OSQP wrapper) on the host.  Here the MOI buffers stay in HBM and
 *     minimize 1/2 x'Px + q'x   subject to   l <= Ax <= u      (P upper triangular; P, A in CSC with 0-based Int64 indices)
 * is rebuilt on the device: the CSC structure depends only on the static indices (pmt_csc_order, host, once), the values are
 * per-run sums of MOI coefficients (pmt_csc_values_f64, per re-evaluation), the bounds come from the constants and the set.
 *   pmt_csc_order: rows/cols are the 1-BASED optimizer indices as they stand in the MOI terms; `upper` folds (i,j) onto
 *     (min,max) (ScalarQuadraticTerm: Q_ij = Q_ji).  Outputs: perm[nnz_in] (terms sorted by column, then row, stable), seg_ptr[nnz_out+1]
 *     (runs of equal (row, col)), col_ptr[ncols+1], row_idx[nnz_out] (0-based), *nnz_out.  Arrays sized for nnz_in suffice.
 *   pmt_csc_values_f64: dst[dst_index ? dst_index[s] : s] = alpha * sum_{p in run s} coeff(perm[p]); src_coeff points at the
 *     coefficient field of term 0 (offset 0 for linear/quadratic terms, 8 for pmt_vector_affine_term), stride = sizeof(term).
 *     dst_index places a block's entries inside a larger matrix (several constraint blocks stacked) or scatters q.
 *   pmt_qp_bounds_f64: row i of `f(x) in set`, f = a'x + c:  l = u = v - c (EQUAL) | l = v - c, u = +infty (GREATER) | l = -infty, u = v - c (LESS).
 * ------------------------------------------------------------------------------------- */
enum { PMT_SET_EQUAL = 0, PMT_SET_GREATER = 1, PMT_SET_LESS = 2 };
int pmt_csc_order(int64_t nnz_in, const int64_t *host_rows, const int64_t *host_cols, int64_t nrows, int64_t ncols, int upper,
                  int64_t *host_perm, int64_t *host_seg_ptr, int64_t *host_col_ptr, int64_t *host_row_idx, int64_t *nnz_out);
int pmt_csc_values_f64(const void *src_coeff, int64_t src_stride_bytes, int64_t nnz_in, const int64_t *perm, const int64_t *seg_ptr,
                       int64_t nnz_out, double alpha, const int64_t *dst_index, double *dst_values, void *stream);
int pmt_qp_bounds_f64(const double *consts, int64_t rows, int set_kind, double set_value, double infty, double *l, double *u,
                      void *stream);
/* Several term buffers feeding one matrix / several constraint blocks in ONE launch each (a model with k constraint blocks otherwise
 * pays 2k small kernels and their in-stream gaps per re-evaluation):
 *   pmt_csc_values_gather_f64: as pmt_csc_values_f64 with term_ptr[p] = device address of the coefficient of the p-th term in CSC order
 *     (the caller folds base pointer, stride and permutation of every block into it, once);
 *   pmt_qp_bounds_rows_f64: row i reads its constant through const_ptr[i] and has its own set kind / value. */
int pmt_csc_values_gather_f64(const double *const *term_ptr, int64_t nnz_in, const int64_t *seg_ptr, int64_t nnz_out, double alpha,
                              const int64_t *dst_index, double *dst_values, void *stream);
/* A DENSE block of a solver matrix: column j of dst (dst + j*dst_pitch doubles) takes the `rows` doubles of column j of src
 * (src + j*src_pitch) — the CSC values of a dense constraint block C*x (+|-) d are the Parameter matrix C itself, column by column,
 * placed at the block's row range of every column of the stacked matrix.  dst_offset != NULL: column j goes to dst + dst_offset[j]
 * instead (the other blocks do not have the same height in every column).  No term structs are read. */
int pmt_copy_2d_f64(const double *src, int64_t src_pitch, double *dst, int64_t dst_pitch, const int64_t *dst_offset, int64_t rows,
                    int64_t cols, void *stream);
int pmt_qp_bounds_rows_f64(const double *const *const_ptr, const int *set_kind, const double *set_value, int64_t rows, double infty,
                           double *l, double *u, void *stream);

/* ---------------------------------------------------------------------------------------
 * Sparse constraint matrix (BASELINE config 5): C given in CSC (Julia SparseMatrixCSC: colptr/rowval
 * 1-based Int64, nzval).  The reference has no sparse path — matvecmul!(y, A::AbstractMatrix, x) src/functions.jl:775-798 walks every
 * (row, col) — so the output here is the reference's output minus the structural zeros.  pmt_sparse_rowmajor_order computes ONCE the
 * row-major order of the structural non-zeros; per re-evaluation pmt_sparse_pack_vector_f64 gathers nzval through it into
 * MOI.VectorAffineTerms (update! src/moi_interop.jl:64-81; row-major, columns ascending within a row = matvecmul!'s order).
 * ------------------------------------------------------------------------------------- */
/* host-side helper: perm[t] = index into nzval of the t-th term in row-major order, rows_out[t], cols_out[t] (1-based) */
int pmt_sparse_rowmajor_order(int64_t m, int64_t n, const int64_t *host_colptr, const int64_t *host_rowval,
                              int64_t *host_perm, int64_t *host_rows, int64_t *host_cols, int64_t *host_row_ptr);
int pmt_sparse_pack_vector_f64(const double *nzval, const int64_t *perm, const int64_t *term_row, const int64_t *term_var,
                               int64_t nnz, const int64_t *varmap, int64_t row_offset,
                               pmt_vector_affine_term *out_terms, void *stream);
/* native form of the same node (Vector{AffineFunction} with ragged rows): out[t] = (nzval[perm[t]], term_var[t]) */
int pmt_sparse_assemble_f64(const double *nzval, const int64_t *perm, const int64_t *term_var, int64_t nnz,
                            pmt_linear_term *out_terms, void *stream);
/* XCD-aware form of the two entry points above (same outputs, bit for bit): columns are cut into `nslab` slabs, workgroup b works on
 * slab b % nslab, so with nslab = 8 (the XCD count) each XCD's L2 only sees one eighth of nzval.  slab_ptr[row*(nslab+1) + s] = index of
 * the first term of `row` whose column is in slab s (host helper, once per pattern; term_col from pmt_sparse_rowmajor_order). */
int pmt_sparse_slab_ptr(int64_t rows, int64_t cols, int nslab, const int64_t *host_row_ptr, const int64_t *host_term_col,
                        int64_t *host_slab_ptr);
int pmt_sparse_pack_vector_slabs_f64(const double *nzval, const int64_t *perm, const int64_t *term_var, const int64_t *slab_ptr,
                                     int64_t rows, int nslab, const int64_t *varmap, int64_t row_offset,
                                     pmt_vector_affine_term *out_terms, void *stream);
int pmt_sparse_assemble_slabs_f64(const double *nzval, const int64_t *perm, const int64_t *term_var, const int64_t *slab_ptr,
                                  int64_t rows, int nslab, pmt_linear_term *out_terms, void *stream);
/* the same two entry points with 32-bit perm / term_var (same values; half the index stream).  Usable when nnz and every variable index
 * (after varmap, if the caller folds it in — see below) are below 2^32.
 * Folding varmap in: pass term_var[t] = varmap[x-variable of term t] and varmap = NULL; the per-term varmap gather disappears (the map
 * only changes when the optimizer's index map does, src/model.jl:100-107, not per re-evaluation). */
int pmt_sparse_pack_vector_slabs_u32_f64(const double *nzval, const uint32_t *perm, const uint32_t *term_var, const int64_t *slab_ptr,
                                         int64_t rows, int nslab, const int64_t *varmap, int64_t row_offset,
                                         pmt_vector_affine_term *out_terms, void *stream);
int pmt_sparse_assemble_slabs_u32_f64(const double *nzval, const uint32_t *perm, const uint32_t *term_var, const int64_t *slab_ptr,
                                      int64_t rows, int nslab, pmt_linear_term *out_terms, void *stream);

/* Block form of the same two nodes (same outputs, bit for bit): the matrix is cut into blocks of 128 rows x `cw` columns; a workgroup reads
 * the coefficients of its block with coalesced loads (the rows of a CSC column ascend, so a column's part of a block is one contiguous run
 * of nzval) into LDS and writes each row's terms of the column band from there.  No per-term gather from HBM / L2 and a 4-byte instead of
 * an 8-byte static index per term: 36 bytes per non-zero.  The variable word of a term comes from the per-COLUMN array col_var[cols]
 * (x[col]; with `varmap` non-null varmap[col_var[col] - 1] as in moi_interop.jl:64-81).  out_consts (optional) = 0.0 (+|-) d[row], the
 * constants of C*x (+|-) d, written by the same launch (d may be NULL with sign 0: zeros).
 * Host helpers, once per pattern:
 *   pmt_sparse_blocks_width  -> *out_cw = the widest band width (power of two, 32..1024) whose blocks all fit the kernel's LDS buffer,
 *                               or 0 when the form does not apply (rows not ascending within a column, 2^32 or more non-zeros, empty matrix):
 *                               use the slab form then;
 *   pmt_sparse_blocks_build  -> desc[ceil(m/128) * n] (8 bytes per (row block, column)), idx[nnz] (4 bytes per term, row-major order),
 *                               band_ptr[m * (ceil(n/cw) + 1)]; perm / term_col / row_ptr from pmt_sparse_rowmajor_order. */
int pmt_sparse_blocks_width(int64_t m, int64_t n, const int64_t *host_colptr, const int64_t *host_rowval, int *out_cw);
int pmt_sparse_blocks_build(int64_t m, int64_t n, const int64_t *host_colptr, const int64_t *host_rowval, const int64_t *host_perm,
                            const int64_t *host_term_col, const int64_t *host_row_ptr, int cw, uint64_t *host_desc, uint32_t *host_idx,
                            int64_t *host_band_ptr);
int pmt_sparse_pack_vector_blocks_f64(const double *nzval, const uint64_t *desc, const uint32_t *idx, const int64_t *band_ptr,
                                      const int64_t *col_var, int64_t rows, int64_t cols, int64_t nnz, int cw, const int64_t *varmap,
                                      int64_t row_offset, const double *d, int sign, pmt_vector_affine_term *out_terms,
                                      double *out_consts, void *stream);
int pmt_sparse_assemble_blocks_f64(const double *nzval, const uint64_t *desc, const uint32_t *idx, const int64_t *band_ptr,
                                   const int64_t *col_var, int64_t rows, int64_t cols, int64_t nnz, int cw, const double *d, int sign,
                                   pmt_linear_term *out_terms, double *out_consts, void *stream);
/* constants of the same node: out[i] = 0.0 (+|-) d[i]  (vecadd!/vecsubtract! on zero!'d functions, src/functions.jl:244,452,474) */
int pmt_consts_f64(const double *d, int64_t n, int sign, double *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Batched independent QPs (BASELINE config 4).  In the reference a batch is many independent Models (src/model.jl:1-22); all
 * instances share one structure, so per re-evaluation only coefficients are produced: one slab of
 * pmt_batch_lsq_slab_doubles(n, m) doubles per instance,
 *     [ Q: n(n+1)/2 | q: n | const: 1 | C: m*n row-major | d-consts: m ],
 * = the coefficients of the canonical MOI objective of residual . residual (Appendix A.3) and of the constraint block
 * C*x (+|-) d (Appendix A.4).  Inputs are instance-major: A[B][r*n] column-major, b[B][r], C[B][m*n] column-major, d[B][m].
 * pmt_batch_expand_f64 rebuilds the full MOI term buffers (with indices through xvar / varmap) of ONE instance from its slab.
 * ------------------------------------------------------------------------------------- */
int64_t pmt_batch_lsq_slab_doubles(int64_t n, int64_t m);
int pmt_batch_lsq_coeffs_f64(const double *A, const double *b, const double *C, const double *d, int64_t B, int64_t n, int64_t r,
                             int64_t m, int sign_b, int sign_d, double *out, int64_t out_stride, void *stream);
int pmt_batch_expand_f64(const double *slab, int64_t n, int64_t m, const int64_t *xvar, const int64_t *varmap,
                         pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                         pmt_vector_affine_term *out_vat, double *out_vconsts, void *stream);

/* ---------------------------------------------------------------------------------------
 * Device-side Parameter update callbacks for synthetic inputs (counter-based):
 * dst[i] = scale * U[0,1)(seed, i)   — the analogue of `Parameter(rand!, zeros(n, n), model)` README.md:36-43
 * ------------------------------------------------------------------------------------- */
int pmt_fill_uniform_f64(double *dst, int64_t n, uint64_t seed, double scale, void *stream);
/* column-major rows x cols matrix with leading dimension lda: dst[c*lda + i] = scale * U(seed, c*rows + i) — same values as the
 * contiguous stream.  Device copies of Parameter matrices whose column stride would be a multiple of 4 KiB are kept with a padded
 * lda: a power-of-two stride puts every column segment of a tile on the same memory channel. */
int pmt_fill_uniform_matrix_f64(double *dst, int64_t rows, int64_t cols, int64_t lda, uint64_t seed, double scale, void *stream);
/* The matrix fill with the seed read from the HOST word *seed_word at every launch — recorded into a plan, at every replay: a recorded
 * Parameter callback (README.md:36-43 rand!, new values at every update!) advances by the host storing the next seed into the word before
 * pmt_plan_update, with no call of its own and, in a small plan, no launch of its own.  cols == 1, lda == rows: a vector.  The word must
 * outlive the plan's tape.  A tape holding such an entry is not captured into a hipGraph. */
int pmt_fill_uniform_dyn_f64(double *dst, int64_t rows, int64_t cols, int64_t lda, const uint64_t *seed_word, double scale, void *stream);
/* the same stream from element index_offset on: dst[i] = scale * U(seed, index_offset + i) (shard of a larger array) */
int pmt_fill_uniform_offset_f64(double *dst, int64_t n, uint64_t seed, uint64_t index_offset, double scale, void *stream);

/* ---------------------------------------------------------------------------------------
 * Per-kernel timing report — the device analogue of `findallocs` (src/debug.jl:4-23), which walks the DAG and
 * reports a cost per node.  While enabled, every launch is bracketed by HIP events on its own stream (never inside
 * a hipGraph capture).  pmt_profile_report synchronises on the recorded events and writes one line per kernel:
 * "<kernel>\t<launches>\t<total_ms>\t<min_ms>\t<max_ms>\n"; returns the untruncated length.
 * ------------------------------------------------------------------------------------- */
int pmt_profile_enable(int on);
/* restrict the bracketing to kernels whose name contains `substring` (NULL or "": all) — two event records per launch are not
 * free (a few microseconds of queue time each), so a benchmark times only the kernel it reports on */
int pmt_profile_filter(const char *substring);
int64_t pmt_profile_report(char *host_buf, size_t cap);
/* Measurement hook for ONE kernel inside a step, without the in-stream gap a HIP-event pair around an in-step launch includes: while
 * `device_words` (2 * workgroups uint64 words in device memory, zeroed by the caller) is set, workgroup w < `workgroups` of the MOI pack
 * kernel (affine_tile_kernel<VAT>, pmt_affine_pack_vector_f64) stores its start / end on the constant-rate device clock (wall_clock64)
 * into words 2w, 2w + 1; the launch ran from the smallest start to the largest end.  NULL or 0 (the default) switches it off: one
 * uniform branch. */
int pmt_profile_kernel_stamps(void *device_words, int64_t workgroups);
/* rate of that clock in kHz (hipDeviceAttributeWallClockRate) */
int pmt_device_clock_khz(int device, int *khz);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU exchange of the batched configuration (BASELINE config 4): one process per GPU, rank g owns the instances
 * [g * per_rank, (g + 1) * per_rank); after the exchange every rank holds every slab (`gathered`: nranks * per_rank slabs of `stride`
 * doubles, global instance order).  The reference has no counterpart (a batch is many independent Models, src/model.jl:1-22).
 * RCCL (librccl.so, dlopen'ed on first use) is driven from inside the library: the host only carries the 128-byte unique id from
 * rank 0 to the others (any launcher: torch.distributed, MPI, a file).  xGMI is point-to-point, so the exchange is the DIRECT schedule
 * — one grouped ncclSend / ncclRecv pair per peer, each over that peer's own link — not a ring.
 *   pmt_comm_unique_id        rank 0: a fresh id (128 bytes)
 *   pmt_comm_init_rank        every rank; nranks == 1 with a NULL id needs no RCCL (nothing to exchange); nranks == 1 WITH an id builds a
 *                             real one-rank RCCL communicator whose exchange is a grouped ncclSend/ncclRecv to itself — the N-rank
 *                             code path on a single GPU
 *   pmt_comm_rccl_calls       ncclSend + ncclRecv calls this communicator has issued so far
 *   pmt_batch_num_chunks / pmt_batch_chunk_range / pmt_batch_gathered_offset
 *                             the chunk schedule, pure host arithmetic: chunk c of a rank = local instances [c * chunk, min(., per_rank))
 *                             (chunk <= 0: one chunk); the slab of global instance i sits at i * stride doubles of `gathered`
 *   pmt_batch_allgather_f64   exchange of slabs that are already computed, chunk by chunk, enqueued on `stream`
 *   pmt_batch_step_f64        pmt_batch_lsq_coeffs_f64 over this rank's instances chunk by chunk on `stream`, chunk c on the wire (the
 *                             communicator's own stream) while chunk c + 1 is computed; `stream` is joined with the last exchange.
 *                             `stream` must be a HIP stream (not a plan's recording handle).
 * ------------------------------------------------------------------------------------- */
int pmt_comm_unique_id(void *out_id_128_bytes);
int pmt_comm_init_rank(int nranks, int rank, const void *unique_id_128_bytes, int device, void **out_comm);
int64_t pmt_comm_rccl_calls(void *comm);
int pmt_comm_destroy(void *comm);
/* the shard of rank `rank`: per_rank = total / nranks instances starting at `first`.  The exchange assumes the SAME per_rank on every
 * rank (the gathered buffer is nranks * per_rank slabs): a batch that does not divide is PMT_DIMENSION_MISMATCH, never an uneven split. */
int pmt_batch_shard(int64_t total, int nranks, int rank, int64_t *per_rank, int64_t *first);
int64_t pmt_batch_num_chunks(int64_t per_rank, int64_t chunk);
int pmt_batch_chunk_range(int64_t per_rank, int64_t chunk, int64_t c, int64_t *lo, int64_t *hi);
int64_t pmt_batch_gathered_offset(int rank, int64_t per_rank, int64_t local_instance, int64_t stride);
int pmt_batch_allgather_f64(void *comm, const double *local, double *gathered, int64_t per_rank, int64_t stride, int64_t chunk, void *stream);
int pmt_batch_step_f64(void *comm, const double *A, const double *b, const double *C, const double *d, int64_t per_rank, int64_t n, int64_t r,
                       int64_t m, int sign_b, int sign_d, double *local, double *gathered, int64_t stride, int64_t chunk, void *stream);

/* ---------------------------------------------------------------------------------------
 * Plan = a recorded re-evaluation: device buffers + a tape of the launches above.
 * Built once by the host from the lazy-expression DAG (↔ `dest = deepcopy(expr())` pre-allocation,
 * src/lazyexpression.jl:202,230,243), replayed by every update!(model) (src/model.jl:132-143) with no
 * allocation and no host logic (↔ FunctionWrapper hop per node, FunctionWrappersQuickFix.jl:108-126).
 * ------------------------------------------------------------------------------------- */
typedef struct pmt_plan pmt_plan;

int pmt_plan_create(int device, void *stream /* NULL: plan creates its own stream */, pmt_plan **out);
int pmt_plan_destroy(pmt_plan *plan);
void *pmt_plan_stream(pmt_plan *plan);
/* device allocation owned by the plan (zero-filled); freed by pmt_plan_destroy */
int pmt_plan_alloc(pmt_plan *plan, size_t bytes, void **out_device_ptr);
size_t pmt_plan_bytes_allocated(const pmt_plan *plan);
/* page-locked host memory (e.g. for the MOI function buffers pmt_plan_fetch writes into); not tied to a plan */
int pmt_host_alloc(size_t bytes, void **out_host_ptr);
int pmt_host_free(void *host_ptr);
/* asynchronous copies on the plan's stream */
int pmt_plan_upload(pmt_plan *plan, void *device_dst, const void *host_src, size_t bytes);
int pmt_plan_fetch(pmt_plan *plan, void *host_dst, const void *device_src, size_t bytes);
/* zero a device buffer (setup time): the padding rows of a Parameter matrix / vector must be zero when a kernel is given the padded row count */
int pmt_plan_zero(pmt_plan *plan, void *device_dst, size_t bytes);
/* pitched variants (hipMemcpy2DAsync): `height` rows of `width_bytes`, e.g. the columns of a matrix whose device copy is padded */
int pmt_plan_upload_2d(pmt_plan *plan, void *device_dst, size_t dst_pitch, const void *host_src, size_t src_pitch, size_t width_bytes,
                       size_t height);
int pmt_plan_fetch_2d(pmt_plan *plan, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes,
                      size_t height);
int pmt_plan_synchronize(pmt_plan *plan);
/* The plan's device-side error state for callers that wait for its stream by other means (hipStreamSynchronize, an event, a wait of
 * their own): PMT_OK, or PMT_HIP_ERROR (cleared by the call) when a grid barrier of a fused run timed out during a replay since the last
 * check — the outputs of that re-evaluation are invalid.  pmt_plan_synchronize and pmt_plan_fetch_synchronize check it themselves. */
int pmt_plan_check(pmt_plan *plan);
/* Recorded fetch (while recording; lane-aware like every recorded call): at replay the D2H copy goes to the plan's FETCH stream, ordered
 * behind everything recorded before it on its lane, and runs while the rest of the tape is still busy — results reach a HOST solver
 * (MOI.set, src/moi_interop.jl:134,171) without waiting for the end of the re-evaluation.  host_dst should be page-locked.  The plan's
 * stream does not wait for these copies; pmt_plan_fetch_synchronize blocks the host until all of them (and a delivery of
 * pmt_quad_gram_csc_deliver_f64) have landed; the next pmt_plan_update waits on the device until they have read their buffers.
 * A plan with recorded fetches cannot be instantiated as a hipGraph. */
int pmt_plan_record_fetch(pmt_plan *plan, void *host_dst, const void *device_src, size_t bytes);
/* pitched recorded fetch: `height` rows of `width_bytes`, row r at device_src + r*src_pitch -> host_dst + r*dst_pitch.  The CSC values of a
 * DENSE constraint block are its Parameter matrix column by column (update!(::MOI.VectorAffineFunction) writes one term per entry,
 * src/moi_interop.jl:64-81): they leave straight out of the Parameter's (padded) device buffer for the block's row range in every column
 * of the solver's stacked matrix — no kernel, the copy engine's rectangle copy where it exists. */
int pmt_plan_record_fetch_2d(pmt_plan *plan, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes,
                             size_t height);
/* HOST -> HOST pitched copy on `threads` worker threads (0: eight; small copies use fewer): `height` rows of `width_bytes`.  For the dense
 * blocks of a host solver's A whose Parameter the HOST updates (`Parameter(model, val=buf)`, src/parameter.jl:88): their values are already
 * on the host and are copied there — from the Parameter's buffer into the block's row range of every column of the solver's array — while
 * the device re-evaluates the rest, instead of coming back over PCIe.  Synchronous; no device work; the worker threads are created and joined
 * inside the call (the library keeps no threads of its own). */
int pmt_host_copy_2d(void *host_dst, size_t dst_pitch, const void *host_src, size_t src_pitch, size_t width_bytes, size_t height, int threads);
/* host: block until the plan's recorded fetches / deliveries have landed; errors as pmt_fetch_synchronize */
int pmt_plan_fetch_synchronize(pmt_plan *plan);

/* Staged (overlapped) uploads of host-updated Parameter values — `Parameter(model, val=buf)` / `Parameter(f, val, model)`,
 * src/parameter.jl:57,88,101-102: the user (or f) rewrites a host buffer between solves and update!() reads it when the Parameter is
 * evaluated.  pmt_plan_upload puts that PCIe copy on the plan's stream, serially in front of the kernels.  A staged upload goes through
 * the plan's COPY stream into a second device buffer (`device_staging`, same size / pitch as the Parameter's buffer) and therefore runs
 * while the kernels of the previous re-evaluation are still busy; the next update consumes it on the plan's stream:
 *     pmt_plan_stage_upload[_2d]   copy stream: waits until the previous commit has read the staging buffer, H2D, records "staged"
 *     pmt_plan_wait_staged         plan stream: waits for every staged upload issued so far
 *     pmt_plan_commit_staged       pmt_plan_wait_staged + device-to-device copy staging -> Parameter buffer (a Parameter that needs a
 *                                  transposition / permutation anyway runs that kernel from the staging buffer instead)
 *     pmt_plan_staging_consumed    plan stream: records "consumed" behind the commits (call once after the last commit of an update)
 *     pmt_plan_staged_synchronize  host: blocks until the staged uploads have left the HOST buffers (which may then be overwritten)
 * The host buffers must be page-locked (pmt_host_alloc) for the copy to be asynchronous. */
int pmt_plan_stage_upload(pmt_plan *plan, void *device_staging, const void *host_src, size_t bytes);
int pmt_plan_stage_upload_2d(pmt_plan *plan, void *device_staging, size_t dst_pitch, const void *host_src, size_t src_pitch, size_t width_bytes,
                             size_t height);
int pmt_plan_wait_staged(pmt_plan *plan);
int pmt_plan_commit_staged(pmt_plan *plan, void *device_dst, const void *device_staging, size_t bytes);
int pmt_plan_staging_consumed(pmt_plan *plan);
/* Lane of the commits that follow (pmt_plan_wait_staged / pmt_plan_commit_staged): 0 = the plan's stream (default), 1 = its side stream, the
 * one the side-lane entries of the tape run on (pmt_plan_set_lane).  A Parameter that ONLY side-lane entries read — the data of a constraint
 * whose MOI copy sits on the side lane — may be committed there: the plan's stream, i.e. the contraction of a least-squares objective, then
 * starts at once instead of waiting for this solve's upload (config 3 through the host hand-off: -0.3 ms per solve).  The caller
 * guarantees that nothing on the plan's stream reads those Parameters; pmt_plan_staging_consumed covers both lanes. */
int pmt_plan_commit_lane(pmt_plan *plan, int lane);
int pmt_plan_staged_synchronize(pmt_plan *plan);
/* Two staging SLOTS (0 / 1): the four calls above act on the current slot, each slot with its own staged / consumed events.  Alternating
 * the slot — and the staging buffers — from one update to the next lets the copy of update k+1 start while the commits of update k are
 * still reading theirs (with one slot it has to wait for them).  Optional: without this call everything uses slot 0. */
int pmt_plan_stage_slot(pmt_plan *plan, int slot);

/* recording: between begin_record and end_record every pmt_*_f64 call issued with stream ==
 * pmt_plan_recording_stream(plan) is appended to the plan's tape instead of being launched */
int pmt_plan_begin_record(pmt_plan *plan);
int pmt_plan_end_record(pmt_plan *plan);
/* While recording: lane 1 marks the following calls as SIDE-LANE entries, lane 0 (the default) returns to the plan's stream.  A side-lane
 * entry must read only buffers that are complete before pmt_plan_update starts (Parameter values) and write outputs that no other
 * entry of the tape reads — the MOI copy of a constraint built straight from its Parameters (update! of one Constraint,
 * src/moi_interop.jl:168-175, is independent of every other record).  At replay these entries fork from the plan's stream at the
 * top of the tape and join at its end; they are queued on the stream's side stream BEHIND the two small reductions of a canonical
 * least-squares objective, so they are dispatched when the contraction's workgroups are already placed and run while those drain and
 * while the fix-up pass runs, instead of adding their own kernels and in-stream gaps behind it.
 * Lane 2 is the FRONT of the side lane: the same stream, but replayed before every other entry of the tape whatever its position in the
 * recording — for an entry that needs nothing of this re-evaluation, e.g. the recorded fetch of a dense constraint block's CSC values
 * straight out of its Parameter buffer (pmt_plan_record_fetch_2d): PCIe is busy from the first microseconds of the solve.
 * Lane 3 is lane 2 WITHOUT the fork from the plan's stream: for an entry whose inputs are produced on the side stream itself (a Parameter
 * committed there, pmt_plan_commit_lane, or regenerated there, pmt_plan_lane_stream) — it does not wait for what the plan's stream still has
 * to do in front of the tape (the callbacks of the objective's Parameters).  The caller vouches for that; everything joins at the end. */
int pmt_plan_set_lane(pmt_plan *plan, int lane);
/* The HIP stream the entries of a lane are replayed on (0: pmt_plan_stream; 1, 2: the side stream).  A device-side Parameter callback whose
 * value only side-lane entries read may run THERE (in front of pmt_plan_update): a transfer at the front of the side lane then does not
 * wait for the callbacks of the objective's Parameters on the plan's stream.  Work issued on it is ordered before the side-lane entries
 * of the next update and joined into the plan's stream at that update's end. */
int pmt_plan_lane_stream(pmt_plan *plan, int lane, void **out_stream);
void *pmt_plan_recording_stream(pmt_plan *plan);
int64_t pmt_plan_tape_length(const pmt_plan *plan);
/* SMALL PLANS.  The reference walks a model's lazy-expression DAG at nanoseconds per hop (src/lazyexpression.jl:50-61; README Example 1
 * re-evaluates in ~15 us on a CPU core, README.md:132-136); a tape replay pays one kernel launch (~5 us) per hop whatever its size.
 * pmt_plan_end_record therefore replaces every run of two or more consecutive SMALL entries of the tape (lane 0; pmt_fill_uniform_*,
 * pmt_affine_assemble_f64, pmt_affine_pack_vector_f64, pmt_quad_expand_f64, pmt_vars_addsub_f64, pmt_consts_f64, pmt_pack_*_f64,
 * pmt_copy_bytes, pmt_transpose_f64, pmt_affvec_combine_f64 (uniform rows), pmt_affvec_scale_f64, pmt_matvecmul_affs_f64, the pmt_vecdot_*
 * forms, pmt_bilinear_f64, pmt_quad_combine_f64, pmt_quad_scale_f64, pmt_scale_*_f64, and pmt_quad_gram_f64 for tiny shapes — each writing
 * at most 32768 elements, a run at most 65536) by ONE launch of an interpreter kernel that executes the
 * run's nodes in tape order with a barrier between dependent ones (csrc/small.hip; one workgroup, or up to 32 with a grid barrier for
 * runs of tens of thousands of elements: pmt_plan_fused_workgroups).  Every element is computed by the same expression
 * as in the entry's own kernel: outputs are bit-identical.  Automatic; larger entries and everything else replay as recorded.
 *   pmt_plan_set_fusion  0: replay the tape as recorded (A/B and tests); 1 (default): fuse.  Not while recording / after graph capture.
 *   pmt_plan_fused       number of fused runs, tape entries they replace, and launches-or-entries one replay executes */
int pmt_plan_set_fusion(pmt_plan *plan, int on);
int pmt_plan_fused(const pmt_plan *plan, int *groups, int *nodes, int64_t *exec_length);
/* Inside a fused run a workgroup barrier stands only in front of a node that touches what an earlier node since the last barrier wrote
 * (or writes what one read): independent nodes — the Parameter callbacks; the objective's chain and a constraint's — share a PHASE.
 * Number of phases over all fused runs (README Example 1: 7 entries, 3 phases). */
int pmt_plan_fused_phases(const pmt_plan *plan);
/* workgroups of the plan's largest fused run (1 .. 32: one per 4096 elements of work beyond 8192, never more than HALF of what the CUs
 * of the plan's stream hold at once — occupancy calculator x the stream's CU mask / the partition's CU count; the barrier in front of a
 * dependent node is then a grid barrier on a counter the plan owns; a hipGraph replays every run with ONE workgroup).  At most one such
 * run is in flight per device: while another plan's is, a launch goes out on one workgroup (same results).  A grid barrier that still
 * times out (kernels of other processes holding the CUs) is reported by pmt_plan_synchronize / pmt_plan_fetch_synchronize / pmt_plan_check. */
int pmt_plan_fused_workgroups(const pmt_plan *plan);
/* update!(m::Model) of a SMALL model behind ONE call (csrc/modelrun.hip) — what src/model.jl:132-143 does per solve, for a host that
 * has recorded its model's tape in the small-model form:
 *   - setdirty! + the Parameter refresh (src/parameter.jl:93-104): a HOST-updated Parameter's value array is copied into its page-locked
 *     mailbox in the device layout (the tape's first entries copy mailbox -> Parameter buffer inside the one launch); a DEVICE-regenerated
 *     Parameter's seed word is advanced (word = base + stride * number of updates so far);
 *   - the re-evaluation + MOI copies (src/model.jl:134-143, src/moi_interop.jl:131-137,168-175): pmt_plan_update;
 *   - results: registered fetches (pmt_model_add_fetch) are enqueued behind the replay; with synchronize != 0 the call returns when the
 *     plan's stream is idle (pmt_plan_synchronize, incl. the fused runs' barrier error) and stores the registered constants
 *     (*dst = *src, e.g. MOI function .constant fields); with 0, pmt_model_wait does that later.
 * A pmt_model owns nothing: plan, mailboxes, seed words, host arrays stay the caller's and must outlive it.
 *   pmt_model_add_mailbox   `host`: rows x cols doubles with strides (row_stride, col_stride) in doubles — Julia Matrix: (1, size(A, 1)),
 *                           numpy C order: (shape[1], 1); cols == 0: a vector / scalar.  `mailbox`: `ld` doubles per column.  *out_slot = the
 *                           slot's index (registration order) = its byte in pmt_model_update's dirty mask
 *   pmt_model_set_host      the value array moved (an out-of-place callback returned a new array)
 *   pmt_model_update        dirty: one byte per slot, or NULL = every slot changed (setdirty!(model) + callbacks that always write).
 *                           A mailbox is only rewritten after the previous replay has been waited for (the launch reads it). */
typedef struct pmt_model pmt_model;
int pmt_model_create(pmt_plan *plan, pmt_model **out);
int pmt_model_destroy(pmt_model *model);
int pmt_model_add_mailbox(pmt_model *model, const double *host, int64_t rows, int64_t cols, int64_t row_stride, int64_t col_stride,
                          double *mailbox, int64_t ld, int *out_slot);
int pmt_model_add_seed(pmt_model *model, uint64_t *seed_word, uint64_t base, uint64_t stride, int *out_slot);
int pmt_model_set_host(pmt_model *model, int slot, const double *host);
int pmt_model_add_constant(pmt_model *model, const double *src, double *dst);
int pmt_model_add_fetch(pmt_model *model, void *host_dst, const void *device_src, size_t bytes);
int pmt_model_num_slots(const pmt_model *model);
int pmt_model_update(pmt_model *model, const unsigned char *dirty, int nslots, int synchronize);
int pmt_model_wait(pmt_model *model);
/* replay the tape on the plan's stream: one update!(m::Model) (src/model.jl:132-143) — the loop over FunctionWrapper calls
 * (src/FunctionWrappersQuickFix.jl:108-126) becomes a loop over recorded launches; after pmt_plan_instantiate_graph, one hipGraph launch */
int pmt_plan_update(pmt_plan *plan);
int pmt_plan_instantiate_graph(pmt_plan *plan);

#ifdef __cplusplus
}
#endif
#endif
