// Canonical least-squares objective: canonicalize!(residual . residual) for residual = A*x (+|-) b, fused with
// the MOI copy.  out_quad = upper triangle (row-major) of 2*A'A as QuadraticTerms, out_lin = 2*A'c, out_const = c'c
// (SURVEY.md Appendix A.3).  The only arithmetic-heavy node of the path: r*n*(n+1) fp64 FLOP -> f64 MFMA.
//
// Reference semantics replaced: _vecdot!/muladd! literal expansion (src/functions.jl:702-709,548-576) followed by
// canonicalize! (src/functions.jl:381-386, sort_and_combine! src/util.jl:9-26) and
// update!(::MOI.ScalarQuadraticFunction, ...) (src/moi_interop.jl:45-62).  The reference sums duplicates in
// QuickSort order; here the contraction index runs in row order inside v_mfma_f64_4x4x4_4b_f64 — coefficients
// agree to rounding (tests: <= 1e-12 relative), indices exactly.
//
// The contraction itself is the persistent stream-K kernel of gram_sk.hip; this file holds the node: validation, the two small
// reductions of the node (q = 2 A'c, c'c) on a side stream, and the C-ABI entry points.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace pmt {

int check_strictly_increasing(const int64_t *xvar_dev, int64_t n, void *stream);
int launch_seq_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *out, hipStream_t s);
size_t gram_sk_workspace_bytes(int64_t rows, int64_t cols);
int launch_gram_sk(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *varmap, int moi,
                   pmt_quadratic_term *out_quad, double *out_csc, double alpha, void *workspace, hipStream_t s);

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int GT = 128;          // output tile edge of the contraction (gram_sk.hip)

// out_lin[j] = (2 * sum_i c_i * A[i,j], vm[xvar[j]]),  c_i = 0.0 (+|-) b[i]; one wave per column (coalesced along i)
__global__ __launch_bounds__(256) void gram_linear_kernel(const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                                          const int64_t *__restrict__ xvar, const double *__restrict__ b, int sign,
                                                          int moi, const int64_t *__restrict__ varmap, LT *__restrict__ out_lin) {
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const int lane = threadIdx.x & 63;
    const double *a = A + col * lda;
    double acc = 0.0;
    if (b && sign)
        for (int64_t i = lane; i < rows; i += 64) acc += signed_const(b[i], sign) * a[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) {
        LT t;
        t.coeff = 2 * acc;
        const int64_t v = xvar[col];
        t.var = moi ? map_var(varmap, v) : v;
        out_lin[col] = t;
    }
}

}  // namespace pmt

namespace pmt {

// One non-blocking side stream + fork/join events PER CALLING STREAM (= per plan: a plan is one stream), created on first use on the
// calling stream's device.  Two plans driven from two host threads therefore never share an event (different plans are
// independent); calls on ONE stream must be serialised by the caller, as for any HIP stream.
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; int device = -1; };
static std::mutex g_side_mu;
static std::unordered_map<hipStream_t, SideStream> g_side;
static SideStream *side_stream(hipStream_t s) {
#ifdef PMT_TUNING
    static const bool enabled = [] { const char *e = getenv("PMT_GRAM_SIDE_STREAM"); return !(e && e[0] == '0'); }();
    if (!enabled) return nullptr;
#endif
    int dev = 0;
    if (hipStreamGetDevice(s, &dev) != hipSuccess) {
        (void)hipGetLastError();
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> lock(g_side_mu);
    SideStream &ss = g_side[s];
    if (ss.stream && ss.device == dev) return &ss;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != dev && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    bool ok = hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
    if (prev != dev) (void)hipSetDevice(prev);
    if (!ok) { (void)hipGetLastError(); g_side.erase(s); return nullptr; }
    ss.device = dev;
    return &ss;
}

}  // namespace pmt

using namespace pmt;

extern "C" size_t pmt_quad_gram_workspace_bytes(int64_t rows, int64_t cols) {
    return gram_sk_workspace_bytes(rows, cols);
}

// the whole node: contraction on the main stream, q = 2A'c and c'c on a side stream.  out_quad (term structs) and out_csc (solver
// values, alpha-scaled) are independent optional outputs of the same contraction.
static int gram_node(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                     int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                     double *out_const, void *workspace, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "quad_gram: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "quad_gram: lda < rows");
    PMT_REQUIRE(sign >= -1 && sign <= 1, PMT_INVALID_ARGUMENT, "quad_gram: sign must be -1, 0 or +1");
    PMT_REQUIRE(out_const, PMT_INVALID_ARGUMENT, "quad_gram: null out_const");
    PMT_REQUIRE(sign == 0 || b || rows == 0, PMT_INVALID_ARGUMENT, "quad_gram: sign != 0 needs b");
    if (cols > 0) PMT_REQUIRE(xvar && (out_quad || out_csc) && out_lin && (A || rows == 0), PMT_INVALID_ARGUMENT, "quad_gram: null pointer");
    PMT_REQUIRE(cols < (int64_t)GT * 46000, PMT_DIMENSION_MISMATCH, "quad_gram: too many columns");
    if (int rc = check_strictly_increasing(xvar, cols, stream)) return rc;
    return dispatch(stream, [=](hipStream_t s) {
        // fork: the two small reductions of this node (q = 2 A'c, HBM-bound; c'c, a serial chain) run on a side stream while
        // the MFMA-bound contraction owns the main stream; join before returning control of `s`.  Legal under stream capture.
        SideStream *side = side_stream(s);
        hipStream_t s2 = s;
        if (side) {
            PMT_HIP_CHECK(hipEventRecord(side->fork, s));
            PMT_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
            s2 = side->stream;
        }
        int rc = PMT_OK;
#ifdef PMT_GRAM_NO_SIDE_KERNELS                          // ablation builds only (tools/): what the co-resident reductions cost the contraction
        if (false) {
#else
        if (cols > 0) {
#endif
            PMT_LAUNCH(gram_linear_kernel, dim3((unsigned)cdiv(cols, 4)), dim3(256), 0, s2, A, lda, rows, cols, xvar, b, sign, moi, varmap, out_lin);
            rc = check_launch("gram_linear_kernel");
        }
        if (!rc) {
            if (b && sign && rows > 0) rc = launch_seq_dot(b, sign, b, sign, rows, out_const, s2);
            else if (hipMemsetAsync(out_const, 0, sizeof(double), s2) != hipSuccess) rc = fail(PMT_HIP_ERROR, "hipMemsetAsync(out_const)");
        }
        if (side) PMT_HIP_CHECK(hipEventRecord(side->join, side->stream));
        if (!rc && cols > 0) rc = launch_gram_sk(A, lda, rows, cols, xvar, varmap, moi, out_quad, out_csc, alpha, workspace, s);
        if (side) PMT_HIP_CHECK(hipStreamWaitEvent(s, side->join, 0));
        return rc;
    });
}

extern "C" int pmt_quad_gram_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                 int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                                 void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_quad, PMT_INVALID_ARGUMENT, "quad_gram: null out_quad");
    return gram_node(A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, nullptr, 1.0, out_lin, out_const, workspace, stream);
}

extern "C" int pmt_quad_gram_csc_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                     const int64_t *varmap, double alpha, double *out_P_values, pmt_quadratic_term *out_quad,
                                     pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_P_values, PMT_INVALID_ARGUMENT, "quad_gram_csc: null out_P_values");
    return gram_node(A, lda, rows, cols, xvar, b, sign, 1, varmap, out_quad, out_P_values, alpha, out_lin, out_const, workspace, stream);
}
