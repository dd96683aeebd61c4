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
int launch_blocked_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *scratch, double *out, hipStream_t s);
size_t blocked_dot_scratch_doubles();
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

// Tall matrices (rows >> cols): one wave per column leaves most of the chip idle (cols = 128: 128 waves read 1 GB) — the rows are cut
// into `nsplit` chunks, one wave per (column, chunk), and the chunk sums of a column are added in chunk order (deterministic).
__global__ __launch_bounds__(256) void gram_linear_split_kernel(const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                                                const double *__restrict__ b, int sign, int64_t chunk,
                                                                double *__restrict__ partial) {
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const int lane = threadIdx.x & 63;
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int len = (int)(min(rows, i0 + chunk) - i0);                   // 32-bit loop state: 16 VGPRs, co-resident with the contraction
    const double *a = A + col * lda + i0, *bb = b + i0;
    double acc = 0.0;
    for (int i = lane; i < len; i += 64) acc += signed_const(bb[i], sign) * a[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) partial[(int64_t)blockIdx.y * cols + col] = acc;
}

__global__ __launch_bounds__(256) void gram_linear_finish_kernel(const double *__restrict__ partial, int nsplit, int64_t cols,
                                                                 const int64_t *__restrict__ xvar, int moi, const int64_t *__restrict__ varmap,
                                                                 LT *__restrict__ out_lin) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    double acc = 0.0;
    for (int k = 0; k < nsplit; ++k) acc += partial[(int64_t)k * cols + col];
    LT t;
    t.coeff = 2 * acc;
    const int64_t v = xvar[col];
    t.var = moi ? map_var(varmap, v) : v;
    out_lin[col] = t;
}

constexpr int64_t LIN_CHUNK_MIN = 4096;      // rows per (column, chunk) wave at least: 64 iterations of 64 lanes
constexpr int64_t LIN_WAVES = 8192;          // waves that fill the chip (256 CUs x 32)
static int linear_splits(int64_t rows, int64_t cols) {
    if (cols <= 0) return 1;
    return (int)std::max<int64_t>(1, std::min(cdiv(LIN_WAVES, cols), rows / LIN_CHUNK_MIN));
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
    // LOWEST priority: (1) HIP multiplexes streams onto a few hardware queues per priority class, so a side stream of its own class
    // never shares a queue with the (normal-priority) stream it serves — sharing one makes the contraction queue up behind its own side
    // kernels (config 2 under torch.distributed, whose RCCL streams take queues too: 1.35 instead of 1.24 ms per step); (2) when both
    // have packets ready, the contraction's workgroups are placed first.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    bool ok = hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, prio_least) == hipSuccess &&
              hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
    if (prev != dev) (void)hipSetDevice(prev);
    if (!ok) { (void)hipGetLastError(); g_side.erase(s); return nullptr; }
    ss.device = dev;
    return &ss;
}

// the side stream of calling stream `s` for other users (the plan's side lane, plan.hip): work queued here lines up BEHIND the Gram
// node's two small reductions, i.e. it is dispatched once the contraction's workgroups are placed and runs as they drain
// a plan that goes away takes the side stream of its stream with it (pmt_plan_destroy): HIP multiplexes streams onto a handful of hardware
// queues, and a leaked side stream can end up sharing the queue of a later plan's stream — its contraction then queues BEHIND its own side
// kernels instead of running beside them (measured: config 3 1.27 -> 1.45 ms when run after another plan in the same process)
void release_side_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it == g_side.end()) return;
    if (it->second.stream) {
        (void)hipStreamSynchronize(it->second.stream);
        (void)hipStreamDestroy(it->second.stream);
    }
    if (it->second.fork) (void)hipEventDestroy(it->second.fork);
    if (it->second.join) (void)hipEventDestroy(it->second.join);
    g_side.erase(it);
}

hipStream_t side_stream_of(hipStream_t s) {
    SideStream *ss = side_stream(s);
    return ss ? ss->stream : nullptr;
}

}  // namespace pmt

using namespace pmt;

extern "C" size_t pmt_quad_gram_workspace_bytes(int64_t rows, int64_t cols) {
    // behind the contraction's partial tiles: chunk sums of q (tall matrices) and the chains of the constant (long vectors)
    return gram_sk_workspace_bytes(rows, cols) + sizeof(double) * ((size_t)linear_splits(rows, cols) * (size_t)std::max<int64_t>(cols, 0) +
                                                                   blocked_dot_scratch_doubles());
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
        double *scratch = workspace ? reinterpret_cast<double *>(static_cast<char *>(workspace) + gram_sk_workspace_bytes(rows, cols)) : nullptr;
        const int nsplit = (scratch && b && sign) ? linear_splits(rows, cols) : 1;
        if (cols > 0 && nsplit > 1) {
            const int64_t chunk = 64 * cdiv(cdiv(rows, nsplit), 64);
            PMT_LAUNCH(gram_linear_split_kernel, dim3((unsigned)cdiv(cols, 4), (unsigned)nsplit), dim3(256), 0, s2, A, lda, rows, cols, b, sign, chunk, scratch);
            PMT_LAUNCH(gram_linear_finish_kernel, dim3((unsigned)cdiv(cols, 256)), dim3(256), 0, s2, scratch, nsplit, cols, xvar, moi, varmap, out_lin);
            rc = check_launch("gram_linear_split_kernel");
        } else if (cols > 0) {
            PMT_LAUNCH(gram_linear_kernel, dim3((unsigned)cdiv(cols, 4)), dim3(256), 0, s2, A, lda, rows, cols, xvar, b, sign, moi, varmap, out_lin);
            rc = check_launch("gram_linear_kernel");
        }
        if (!rc) {
            double *chains = scratch ? scratch + (size_t)linear_splits(rows, cols) * (size_t)cols : nullptr;
            if (b && sign && rows > 0) rc = launch_blocked_dot(b, sign, b, sign, rows, chains, out_const, s2);
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
