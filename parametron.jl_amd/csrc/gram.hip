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

#include "gram_common.h"

namespace pmt {

int check_strictly_increasing(const int64_t *xvar_dev, int64_t n, void *stream);
int launch_blocked_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *scratch, double *out, hipStream_t s);
size_t blocked_dot_scratch_doubles();
size_t gram_sk_workspace_bytes(int64_t rows, int64_t cols);
int launch_courier(const double *src, double *dst_dev, unsigned long long *progress, unsigned *done, int *error, int ngroups,
                   const unsigned long long *expect, const int64_t *off, hipStream_t s);
int launch_to_host(const void *src, void *dst_dev, size_t bytes, hipStream_t s);
void *host_device_pointer(void *host);
struct SKDeliver;
int launch_gram_sk(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *varmap, int moi,
                   pmt_quadratic_term *out_quad, double *out_csc, double alpha, void *workspace, int order_w, const SKDeliver *deliver,
                   hipStream_t s);

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
// `counters` (library-owned device memory, zeroed once here, put back to zero by the courier kernel): the progress counts of a host
// delivery (MAXGROUPS x u64) and the courier's own completion count / error flag — calls on one stream are serialised, so one set per
// calling stream is enough.
// `fetch` is the calling stream's DEVICE-TO-HOST stream (created on first use, highest priority so that it has a hardware queue of its
// own class): recorded fetches (pmt_plan_record_fetch) and the band-wise delivery of pmt_quad_gram_csc_deliver_f64 travel on it while
// the kernels go on; `fetch_done` is recorded behind the last copy enqueued so far.
struct SideStream {
    hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; int device = -1;
    void *counters = nullptr;
    hipStream_t fetch = nullptr; hipEvent_t fetch_done = nullptr; bool fetch_pending = false;
};
// layout of `counters`: [MAXGROUPS x u64 progress][u32 courier done][i32 courier error]
constexpr size_t PROGRESS_OFFSET = 0;
constexpr size_t DONE_OFFSET = PROGRESS_OFFSET + MAXGROUPS * sizeof(unsigned long long);
constexpr size_t COUNTER_BYTES = DONE_OFFSET + 2 * sizeof(unsigned);
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
              hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess &&
              hipMalloc(&ss.counters, COUNTER_BYTES) == hipSuccess &&
              hipMemsetAsync(ss.counters, 0, COUNTER_BYTES, s) == hipSuccess;      // on the calling stream: ordered before its first kernel
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
// Plans that share one external stream share its side stream: it is reference-counted by plan (pmt_plan_create retains, pmt_plan_destroy
// releases) and goes away with the LAST of them, not with the first.
static std::unordered_map<hipStream_t, int> g_side_refs;
void retain_side_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    ++g_side_refs[s];
}
void release_side_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    auto rc = g_side_refs.find(s);
    if (rc != g_side_refs.end()) {
        if (--rc->second > 0) return;
        g_side_refs.erase(rc);
    }
    auto it = g_side.find(s);
    if (it == g_side.end()) return;
    if (it->second.stream) {
        (void)hipStreamSynchronize(it->second.stream);
        (void)hipStreamDestroy(it->second.stream);
    }
    if (it->second.fork) (void)hipEventDestroy(it->second.fork);
    if (it->second.join) (void)hipEventDestroy(it->second.join);
    if (it->second.fetch) {
        (void)hipStreamSynchronize(it->second.fetch);
        (void)hipStreamDestroy(it->second.fetch);
    }
    if (it->second.fetch_done) (void)hipEventDestroy(it->second.fetch_done);
    if (it->second.counters) (void)hipFree(it->second.counters);
    g_side.erase(it);
}

// ---- the calling stream's device-to-host stream -----------------------------------------------------------------------------------
static int ensure_fetch_stream(SideStream *ss) {
    if (ss->fetch) return PMT_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != ss->device) PMT_HIP_CHECK(hipSetDevice(ss->device));
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&ss->fetch, hipStreamNonBlocking, prio_greatest);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ss->fetch_done, hipEventDisableTiming);
    if (prev != ss->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PMT_HIP_ERROR, std::string("fetch stream: ") + hipGetErrorString(e)); }
    return PMT_OK;
}

// D2H copy on the fetch stream of `s`, ordered behind everything enqueued on `after` (the plan's stream or its side stream) so far
int fetch_async(hipStream_t s, hipStream_t after, hipEvent_t order_event, void *host_dst, const void *device_src, size_t bytes) {
    SideStream *ss = side_stream(s);
    if (!ss) return fail(PMT_STATE_ERROR, "fetch_async: no auxiliary streams for this stream");
    if (int rc = ensure_fetch_stream(ss)) return rc;
    PMT_HIP_CHECK(hipEventRecord(order_event, after));
    PMT_HIP_CHECK(hipStreamWaitEvent(ss->fetch, order_event, 0));
    // a <= 16-VGPR copy kernel that is co-resident with the contraction (deliver.hip) when the destination is page-locked, 8-byte words
    // and below 16 GiB; the runtime's copy otherwise
    void *dst_dev = (bytes % 8 == 0 && bytes / 8 < (size_t)1 << 31) ? host_device_pointer(host_dst) : nullptr;
    if (dst_dev) { if (int rc = launch_to_host(device_src, dst_dev, bytes, ss->fetch)) return rc; }
    else PMT_HIP_CHECK(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ss->fetch));
    PMT_HIP_CHECK(hipEventRecord(ss->fetch_done, ss->fetch));
    ss->fetch_pending = true;
    return PMT_OK;
}

// `s` waits until the copies enqueued on its fetch stream so far have read their device buffers (start of the next re-evaluation)
int fetch_fence(hipStream_t s) {
    std::unique_lock<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it == g_side.end() || !it->second.fetch_pending) return PMT_OK;
    SideStream *ss = &it->second;
    lock.unlock();
    PMT_HIP_CHECK(hipStreamWaitEvent(s, ss->fetch_done, 0));
    ss->fetch_pending = false;
    return PMT_OK;
}

// host: block until every copy enqueued on the fetch stream of `s` has landed
int fetch_synchronize(hipStream_t s) {
    std::unique_lock<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it == g_side.end() || !it->second.fetch) return PMT_OK;
    hipStream_t f = it->second.fetch;
    void *counters = it->second.counters;
    lock.unlock();
    PMT_HIP_CHECK(hipStreamSynchronize(f));
    int err = 0;
    PMT_HIP_CHECK(hipMemcpy(&err, static_cast<char *>(counters) + DONE_OFFSET + sizeof(unsigned), sizeof(int), hipMemcpyDeviceToHost));
    if (err) {
        PMT_HIP_CHECK(hipMemset(counters, 0, COUNTER_BYTES));
        return fail(PMT_HIP_ERROR, "host delivery: the courier saw no progress of the contraction for 2 s and gave up; the host arrays are incomplete");
    }
    return PMT_OK;
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

// Host delivery of the CSC values (pmt_quad_gram_csc_deliver_f64): where the band groups end and what each of them ships.
struct DeliverPlan {
    double *host = nullptr;                 // page-locked destination, same layout as out_csc
    double *host_dev = nullptr;             // ... and its device-visible address
    int ngroups = 0;
    short gend[MAXGROUPS];                  // band group i = tile columns [gend[i-1], gend[i])
    unsigned long long expect[MAXGROUPS];   // accumulator units of the group's tiles = 32 x sum over its bands kb of (kb + 1)
    int64_t off[MAXGROUPS + 1];             // CSC offsets (doubles) of the groups' first columns
};

// band groups of (nearly) equal bytes: group i ends at the first band whose columns bring the shipped fraction to (i + 1) / ngroups
static DeliverPlan deliver_plan(int64_t cols, int ngroups, double *host) {
    DeliverPlan d;
    d.host = host;
    const int nt = (int)cdiv(cols, GT);
    const int64_t total = cols * (cols + 1) / 2;
    ngroups = std::max(1, std::min(ngroups, std::min(nt, MAXGROUPS)));
    int g = 0, b0 = 0;
    d.off[0] = 0;
    for (int kb = 0; kb < nt; ++kb) {
        const int64_t cend = std::min<int64_t>(cols, (int64_t)(kb + 1) * GT);
        const int64_t off = cend * (cend + 1) / 2;
        const bool last_band = kb == nt - 1;
        if (last_band || (off * ngroups >= total * (g + 1) && nt - 1 - kb >= ngroups - 1 - g)) {
            d.gend[g] = (short)(kb + 1);
            d.off[g + 1] = off;
            unsigned long long tiles = 0;
            for (int k = b0; k <= kb; ++k) tiles += (unsigned long long)(k + 1);
            d.expect[g] = tiles * 32;           // Cfg<2>::NACC units per tile (gram_sk.hip: sk_signal_tile)
            b0 = kb + 1;
            ++g;
            if (last_band) break;
        }
    }
    d.ngroups = g;
    return d;
}

// the whole node: contraction on the main stream, q = 2A'c and c'c on a side stream.  out_quad (term structs) and out_csc (solver
// values, alpha-scaled) are independent optional outputs of the same contraction.  host_csc != null: out_csc is also DELIVERED to that
// page-locked host buffer, band group by band group, on the stream's fetch stream while the contraction is still running.
static int gram_node(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                     int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                     double *out_const, void *workspace, double *host_csc, int ngroups, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "quad_gram: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "quad_gram: lda < rows");
    PMT_REQUIRE(sign >= -1 && sign <= 1, PMT_INVALID_ARGUMENT, "quad_gram: sign must be -1, 0 or +1");
    PMT_REQUIRE(out_const, PMT_INVALID_ARGUMENT, "quad_gram: null out_const");
    PMT_REQUIRE(sign == 0 || b || rows == 0, PMT_INVALID_ARGUMENT, "quad_gram: sign != 0 needs b");
    if (cols > 0) PMT_REQUIRE(xvar && (out_quad || out_csc) && out_lin && (A || rows == 0), PMT_INVALID_ARGUMENT, "quad_gram: null pointer");
    PMT_REQUIRE(cols < (int64_t)GT * 32000, PMT_DIMENSION_MISMATCH, "quad_gram: too many columns");
    if (int rc = check_strictly_increasing(xvar, cols, stream)) return rc;
    DeliverPlan dplan;
    if (host_csc && cols > 0) {
        dplan = deliver_plan(cols, ngroups > 0 ? ngroups : 8, host_csc);
        dplan.host_dev = static_cast<double *>(host_device_pointer(host_csc));
        PMT_REQUIRE(dplan.host_dev, PMT_INVALID_ARGUMENT, "quad_gram_csc_deliver: host_P_values must be page-locked host memory (pmt_host_alloc)");
    }
    return dispatch(stream, [=](hipStream_t s) {
        // fork: the two small reductions of this node (q = 2 A'c, HBM-bound; c'c, a serial chain) run on a side stream while
        // the MFMA-bound contraction owns the main stream; join before returning control of `s`.  Legal under stream capture.
        SideStream *side = side_stream(s);
        const bool deliver = dplan.host != nullptr;
        if (deliver) {
            PMT_REQUIRE(side && side->counters, PMT_STATE_ERROR, "quad_gram_csc_deliver: no auxiliary streams for this stream");
            if (int rc = ensure_fetch_stream(side)) return rc;
            // the previous delivery must have read out_csc before this contraction overwrites it
            if (side->fetch_pending) { PMT_HIP_CHECK(hipStreamWaitEvent(s, side->fetch_done, 0)); side->fetch_pending = false; }
        }
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
        if (!rc && cols > 0) {
            SKDeliver sd;
            if (deliver) {
                sd.progress = reinterpret_cast<unsigned long long *>(static_cast<char *>(side->counters) + PROGRESS_OFFSET);
                sd.ngroups = dplan.ngroups;
                for (int i = 0; i < MAXGROUPS; ++i) sd.gend[i] = i < dplan.ngroups ? dplan.gend[i] : 0;
            }
            // a delivery wants the column bands finished in ascending order: super-columns of two tile columns
            rc = launch_gram_sk(A, lda, rows, cols, xvar, varmap, moi, out_quad, out_csc, alpha, workspace, deliver ? 2 : 0, deliver ? &sd : nullptr, s);
            if (!rc && deliver) {
                // fetch stream: ONE courier launch, queued now that the contraction's workgroups are on their way; it polls the band groups'
                // counts and stores each finished group straight into the host array (deliver.hip)
                char *cb = static_cast<char *>(side->counters);
                rc = launch_courier(out_csc, dplan.host_dev, sd.progress, reinterpret_cast<unsigned *>(cb + DONE_OFFSET),
                                    reinterpret_cast<int *>(cb + DONE_OFFSET + sizeof(unsigned)), dplan.ngroups, dplan.expect, dplan.off, side->fetch);
                if (rc) return rc;
                PMT_HIP_CHECK(hipEventRecord(side->fetch_done, side->fetch));
                side->fetch_pending = true;
            }
        }
        if (side) PMT_HIP_CHECK(hipStreamWaitEvent(s, side->join, 0));
        return rc;
    });
}

extern "C" int pmt_quad_gram_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                 int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                                 void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_quad, PMT_INVALID_ARGUMENT, "quad_gram: null out_quad");
    return gram_node(A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, nullptr, 1.0, out_lin, out_const, workspace, nullptr, 0, stream);
}

extern "C" int pmt_quad_gram_csc_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                     const int64_t *varmap, double alpha, double *out_P_values, pmt_quadratic_term *out_quad,
                                     pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_P_values, PMT_INVALID_ARGUMENT, "quad_gram_csc: null out_P_values");
    return gram_node(A, lda, rows, cols, xvar, b, sign, 1, varmap, out_quad, out_P_values, alpha, out_lin, out_const, workspace, nullptr, 0, stream);
}

extern "C" int pmt_quad_gram_csc_deliver_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                             const int64_t *varmap, double alpha, double *out_P_values, double *host_P_values, int ngroups,
                                             pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_P_values && host_P_values, PMT_INVALID_ARGUMENT, "quad_gram_csc_deliver: null P values pointer");
    PMT_REQUIRE(ngroups >= 0 && ngroups <= MAXGROUPS, PMT_INVALID_ARGUMENT, "quad_gram_csc_deliver: ngroups must be 0 (default) .. 16");
    return gram_node(A, lda, rows, cols, xvar, b, sign, 1, varmap, nullptr, out_P_values, alpha, out_lin, out_const, workspace, host_P_values, ngroups,
                     stream);
}

extern "C" int pmt_fetch_synchronize(void *stream) {
    PMT_REQUIRE(!is_recording_handle(stream), PMT_INVALID_ARGUMENT, "fetch_synchronize: `stream` is a plan's recording handle; use pmt_plan_fetch_synchronize");
    return fetch_synchronize(reinterpret_cast<hipStream_t>(stream));
}
