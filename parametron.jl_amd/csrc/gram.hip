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

#include <functional>
#include <memory>
#include <vector>

#include "dma.h"
#include "gram_common.h"

#ifndef PMT_GRAM_ABL_NO_LINEAR
#define PMT_GRAM_ABL_NO_LINEAR 0   // ablation (wrong q): the stream-K form without its affine part on the side stream — what folding q into the contraction could gain at most
#endif

namespace pmt {

int check_strictly_increasing(const int64_t *xvar_dev, int64_t n, void *stream);
void mark_no_graph(void *stream);
int launch_blocked_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *scratch, double *out, hipStream_t s, int chained);
int gram_tall_groups(int64_t rows, int64_t cols);
int gram_tall_stage_rows(int64_t rows, int64_t cols);
int gram_tall_run_lanes(int64_t rows, int64_t cols);
size_t blocked_dot_scratch_doubles();
size_t gram_sk_workspace_bytes(int64_t rows, int64_t cols);
int launch_courier(const double *src, double *dst_dev, long long *ready, unsigned *done, int *error, int ngroups, const int64_t *gbeg, const int64_t *gend, hipStream_t s);
int launch_to_host(const void *src, void *dst_dev, size_t bytes, hipStream_t s);
int launch_to_host_2d(const void *src, size_t src_pitch, void *dst_dev, size_t dst_pitch, size_t width_bytes, size_t height, hipStream_t s);
void *host_device_pointer(void *host);
int launch_gram_sk(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *varmap, int moi,
                   pmt_quadratic_term *out_quad, double *out_csc, double alpha, void *workspace, int order_w, int64_t seq_begin, int64_t seq_count,
                   unsigned *pair_flags, unsigned epoch, int *error_word, hipStream_t s, int strict = 0);
bool gram_tall_applies(int64_t rows, int64_t cols);
bool gram_tiny(int64_t rows, int64_t cols);
int launch_small_one(const SmallNode &nd, hipStream_t s);
bool gram_tall_diag_applies(int64_t rows, int64_t cols);
size_t gram_tall_workspace_bytes(int64_t rows, int64_t cols);
int launch_gram_tall(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign, int moi,
                     const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                     double *out_const, void *workspace, hipStream_t s);
size_t gram_mid_workspace_bytes(int64_t rows, int64_t cols);
int gram_mid_counters(int64_t cols);
int launch_gram_mid(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign, int moi,
                    const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                    double *out_const, void *workspace, unsigned *counters, hipStream_t s);
int launch_gram_mid_constant(const double *b, int sign, int64_t rows, double *out_const, hipStream_t s);
constexpr int GT = 128;          // output tile edge of the contraction (gram_sk.hip)
constexpr size_t PAIR_FLAG_BYTES = 4096;      // 4 bytes per tile of a stage (at most 512 workgroups / 2 tiles)
// a CSC delivery computes the tiles column band by column band (super-columns of ONE tile column: a band's completion is never held back by
// its neighbour's, so the stages complete nearly equal byte counts — 1.73 vs 1.77 ms per solve with super-columns of two at n = 4096,
// profiles/r04_host_delivery.txt), walked from the LAST band to the first (order_w < 0): the column bands finish in descending order.  Band kb holds (kb + 1) tiles' worth of values, so the long bands leave while the contraction is
// still busy and what is left to ship when it ends — the tail nothing overlaps — is the short ones (ascending order, round 3, left 8.1 MB
// = 0.15 ms of PCIe behind the last stage at n = 4096; descending leaves 1 MB).
constexpr int DELIVER_ORDER_W = 1;

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
// The constant c'c of the stream-K node: sequential (the reference's left-to-right sum, src/functions.jl:574, bit for bit) or 2048 chains.
// The sequential chain runs on one wave beside the contraction at ~0.07 us per row (0.287 ms at r = 4096, profiles/r04_c2_rocprofv3_kernel_stats.csv);
// the contraction takes T(r) = 0.096 + 0.276 r / 1024 ms at n = 4096 (528 tiles), in proportion to the tile count
// elsewhere.  The chain is kept where the contraction hides it (config 2: 0.29 of 1.19 ms); where it would be half of the node or more
// (few columns) the chained form takes over, as it always does above 8192 rows.  Bit-exactness of the constant is not part of the
// parity bar (1e-12 relative); the order is fixed either way and reported by pmt_quad_gram_constant_order.
static bool constant_chained(int64_t rows, int64_t cols) {
    if (rows > 8192) return true;
    if (rows < 2048) return false;
    const double nt = (double)cdiv(std::max<int64_t>(cols, 1), GT);
    const double contraction_ms = (0.096 + 0.276 * (double)rows / 1024.0) * (nt * (nt + 1) / 2) / 528.0;
    return 0.07e-3 * (double)rows > 0.5 * contraction_ms;
}

// WIDE shapes of up to 2048 columns that the one-launch form on 64 x 64 tiles takes (gram_mid.hip; round 6b/6c): the sizes the reference is
// used at with a few hundred variables, and most of what used to be the four launches tall + fix-up + strict stream-K + fix-up.  Same-box node
// times in us, four launches -> one (profiles/r06_gram_mid.txt): 300 x 300 35.7 -> 16, 40 x 520 46 -> 10, 1024 x 512 44 -> 21, 4096 x 512
// 62 -> 33.6, 4096 x 1024 128 -> 98, 8192 x 512 90 -> 50.5, 2048 x 1280 127 -> 71, 65536 x 512 411 -> 304, 65536 x 1024 1380 -> 1189,
// 262144 x 512 1387 -> 1282, 100000 x 129 188 -> 103.  Up to 128 columns the one-tile kernel of gram_tall.hip stays (three 64 x 64 tiles
// split 80 ways fold too much: 8192 x 128 18 -> 40 us).
#ifndef PMT_MID_MAXCOLS
#define PMT_MID_MAXCOLS 2048
#endif
#ifndef PMT_MID_BIGCOLS
#define PMT_MID_BIGCOLS 4096
#endif
#ifndef PMT_MID_BIGEL
#define PMT_MID_BIGEL ((int64_t)1 << 29)      // (the fast load path needs the matrix within 4 GiB)
#endif
// 2049 .. 4096 columns (config 2), round 6c: with the pinned instruction stream, the tiles in super-tile order and the partial round / the
// diagonal tiles split (gram_mid.hip: mid_plan) the one launch beats the stream-K node (contraction + fix-up, q and the constant on a side
// stream) at every row count measured — 4096 x 4096 1205 -> 1069-1092 us (0.73 -> 0.80-0.82 of the f64 matrix peak), 2048 x 4096 661 -> 573,
// 4096 x 3072 718 -> 630, 4096 x 2304 461 -> 373, 8192 x 2560 1065 -> 839, 1000 x 3000 246 -> 191, 8192 x 4096 2340 -> 2062, 16384 x 4096
// 4616 -> 4405, 32768 x 3072 5338 -> 4863, 131072 x 2560 17877 -> 15256, 20000 x 3500 4878 -> 3859; 65536 x 4096 18287 -> 18444 is a tie.  A
// STAGED host delivery of such a shape (config 2's host_csc hand-off: column bands leave while the contraction runs) keeps the stream-K
// kernel — run_quad_gram — with the constant in this form's order, so that pmt_quad_gram_constant_order holds for every call form.
bool gram_mid_big(int64_t rows, int64_t cols) {
#ifdef PMT_NO_MID
    return false;
#endif
    return cols > 16 * 128 && cols <= PMT_MID_BIGCOLS && rows >= 1 && rows * cols < PMT_MID_BIGEL;
}
bool gram_mid_applies(int64_t rows, int64_t cols) {
#ifdef PMT_NO_MID
    return false;
#endif
    if (gram_mid_big(rows, cols)) return true;
    if (!gram_tall_diag_applies(rows, cols) || cols > PMT_MID_MAXCOLS) return false;
#ifdef PMT_MID_ALWAYS
    return true;                                                    // (A/B builds: every wide shape of up to PMT_MID_MAXCOLS columns)
#endif
    // Measured with the pinned instruction stream of round 6c (gram_mid.hip: mid_step), one launch against four, us (profiles/r06_gram_mid.txt):
    //   wins   200000 x 224 287 (330), 230000 x 256 354 (381), 150000 x 288 304 (459), 160000 x 320 326 (498), 131072 x 384 358 (460),
    //          100000 x 448 375 (554), 262144 x 512 1282 (1387), 524288 x 512 2527 (2680), 2^20 x 384 2991 (3188), 4096 x 2048 359 (381),
    //          16384 x 2048 1190 (1338), 131072 x 1024 2453 (2640), 262144 x 1024 4980 (5240)
    //   loses  380000 x 129 578 (505), 300000 x 160 462 (422), 250000 x 192 389 (366), 262144 x 256 407 (386), 786432 x 320 2317 (2224)
    // (every 64-column panel is read once per tile of its row and column: few, narrow panels out of HBM are the four launches' shapes).  The
    // fast load path needs the matrix within 4 GiB: below 2^29 elements.
    const int64_t el = rows * cols;
    if (cols <= 192) return el <= ((int64_t)1 << 25);
    if (cols < 320) return el < ((int64_t)1 << 26);
    if (cols < 384) return el <= ((int64_t)1 << 27);
    return el < ((int64_t)1 << 29);
}

static int linear_splits(int64_t rows, int64_t cols) {
    if (cols <= 0) return 1;
    return (int)std::max<int64_t>(1, std::min(cdiv(LIN_WAVES, cols), rows / LIN_CHUNK_MIN));
}

}  // namespace pmt

namespace pmt {

// One non-blocking side stream + fork/join events PER CALLING STREAM (= per plan: a plan is one stream), created on first use on the
// calling stream's device.  Two plans driven from two host threads therefore never share an event (different plans are
// independent); calls on ONE stream must be serialised by the caller, as for any HIP stream.
// `counters` (library-owned device memory, zeroed once here, re-armed by the courier kernel): the courier's per-group flags of a host
// delivery (MAXGROUPS x i64) and its own completion count / error flag — calls on one stream are serialised, so one set per
// calling stream is enough.
// `fetch` is the calling stream's DEVICE-TO-HOST stream (created on first use, highest priority so that it has a hardware queue of its
// own class): recorded fetches (pmt_plan_record_fetch) and the band-wise delivery of pmt_quad_gram_csc_deliver_f64 travel on it while
// the kernels go on; `fetch_done` is recorded behind the last copy enqueued so far.
struct SideStream {
    hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr; int device = -1;
    void *counters = nullptr;
    // error word of the kernels that wait on other workgroups with a bound (the courier, the pair fold of gram_sk.hip): page-locked host
    // memory the kernels store to (system scope) and the host reads without a copy in fetch_synchronize.  0 = fine, ERR_* otherwise
    int *err_host = nullptr, *err_dev = nullptr;
    hipStream_t fetch = nullptr; hipEvent_t fetch_done = nullptr; bool fetch_pending = false;
    std::vector<std::pair<dma::Engine *, dma::Signal>> dma_pending;     // completion signals of copy-engine transfers in flight
    std::vector<std::shared_ptr<void>> keepalive;                        // ... and the owners of their signals (an immediate call's go with the call)
    bool in_replay = false;                                              // a plan's tape is being replayed: P's transfers are submitted at its end
    std::vector<std::function<int()>> deferred;
};
constexpr int ERR_COURIER = 1, ERR_PAIR_FOLD = 2;
// layout of `counters`: [MAXGROUPS x u64 unused][MAXGROUPS x i64 courier flags (armed = 1)][u32 courier done][u32 unused]
constexpr size_t PROGRESS_OFFSET = 0;
constexpr size_t FLAGS_OFFSET = PROGRESS_OFFSET + MAXGROUPS * sizeof(unsigned long long);
constexpr size_t DONE_OFFSET = FLAGS_OFFSET + MAXGROUPS * sizeof(long long);
constexpr size_t MID_OFFSET = DONE_OFFSET + 2 * sizeof(unsigned);                  // per-tile arrival counts of the one-launch mid-size node (gram_mid.hip)
constexpr size_t MID_COUNTER_BYTES = 135168;         // 16 words per tile, 2080 tiles at 4096 columns
constexpr size_t COUNTER_BYTES = MID_OFFSET + MID_COUNTER_BYTES;
static std::mutex g_side_mu;
static std::unordered_map<hipStream_t, SideStream> g_side;
static int wait_dma_pending(SideStream *ss);
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
              hipEventCreateWithFlags(&ss.join2, hipEventDisableTiming) == hipSuccess &&
              hipMalloc(&ss.counters, COUNTER_BYTES) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void **>(&ss.err_host), 64, hipHostMallocDefault) == hipSuccess &&
              hipMemsetAsync(ss.counters, 0, COUNTER_BYTES, s) == hipSuccess;      // on the calling stream: ordered before its first kernel
    if (ok) {
        memset(ss.err_host, 0, 64);
        ss.err_dev = static_cast<int *>(host_device_pointer(ss.err_host));
        ok = ss.err_dev != nullptr;
    }
    if (ok) {
        static const long long armed[MAXGROUPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        ok = hipMemcpyAsync(static_cast<char *>(ss.counters) + FLAGS_OFFSET, armed, sizeof armed, hipMemcpyHostToDevice, s) == hipSuccess;
    }
    if (prev != dev) (void)hipSetDevice(prev);
    if (!ok) {
        (void)hipGetLastError();
        if (ss.err_host) (void)hipHostFree(ss.err_host);
        if (ss.counters) (void)hipFree(ss.counters);
        g_side.erase(s);
        return nullptr;
    }
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
    if (it->second.join2) (void)hipEventDestroy(it->second.join2);
    if (it->second.fetch) {
        (void)hipStreamSynchronize(it->second.fetch);
        (void)hipStreamDestroy(it->second.fetch);
    }
    if (it->second.fetch_done) (void)hipEventDestroy(it->second.fetch_done);
    (void)wait_dma_pending(&it->second);
    if (it->second.counters) (void)hipFree(it->second.counters);
    if (it->second.err_host) (void)hipHostFree(it->second.err_host);
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

FetchState::~FetchState() {
    if (eng && created) { dma::signal_destroy(eng, dep); dma::signal_destroy(eng, done); }
}

// D2H copy ordered behind everything enqueued on `after` (the plan's stream or its side stream) so far.  Preferred: the copy engine, started
// by a signal that a one-thread kernel on `after` sets (hsadma.hip) — nothing of it runs on a CU.  Otherwise a kernel copy / the runtime's
// copy on the fetch stream of `s`.  r.height > 0: a PITCHED copy of r.height rows of `bytes` bytes each (a matrix block whose device copy
// is padded, or that lands in a column range of a wider host matrix); the engine does those natively (hsa_amd_memory_async_copy_rect).
int fetch_async(hipStream_t s, hipStream_t after, hipEvent_t order_event, void *host_dst, const void *device_src, size_t bytes, FetchState *st, FetchRect r) {
    SideStream *ss = side_stream(s);
    if (!ss) return fail(PMT_STATE_ERROR, "fetch_async: no auxiliary streams for this stream");
    // A fetch behind the plan's OWN stream inside a replay (its producer finishes with the objective's kernels, e.g. the constant, whose
    // serial chain is itself queued at the end of the replay) is issued at the end of the replay, behind that chain's join and behind the
    // band groups of a delivery (the copy engine's queue is first in, first out).
    if (ss->in_replay && after == s) {
        ss->deferred.push_back([=]() -> int { return fetch_async(s, after, order_event, host_dst, device_src, bytes, st, r); });
        return PMT_OK;
    }
    const int mode = dma::delivery_mode();
    // whichever way this entry's previous copy went, it has read the device buffer (and left the host one) before the next one is queued
    if (st && st->pending) { if (int rc = dma::wait(st->eng, st->done, 10.0)) return rc; st->pending = false; }
    // the engine is handed physical pages: only page-locked, device-mapped destinations qualify (a pageable numpy / Julia array takes the
    // runtime's copy below, which stages it)
    if (st && st->pinned < 0) st->pinned = host_device_pointer(host_dst) ? 1 : 0;
    if (mode != 2 && st && st->pinned == 1 && !st->created && !st->tried) {
        st->tried = true;
        st->eng = dma::get(ss->device);
        if (st->eng) {
            if (dma::signal_create(st->eng, 1, &st->dep) == PMT_OK && dma::signal_create(st->eng, 0, &st->done) == PMT_OK) st->created = true;
            else st->eng = nullptr;
        }
    }
    const bool engine = mode != 2 && st && st->created;
    if (mode == 1 && !engine)
        return fail(PMT_STATE_ERROR, "host delivery: the copy engine was demanded (pmt_set_host_delivery(1)) but is not available for this transfer "
                                     "(no HSA agent match, or a pageable destination)");
    if (engine) {
        dma::signal_set(st->eng, st->dep, 1);
        dma::signal_set(st->eng, st->done, 1);
        if (int rc = dma::launch_signal_store(st->dep, after)) return rc;
        st->pending = true;
        ss->dma_pending.emplace_back(st->eng, st->done);
        if (r.height) return dma::copy_rect_to_host(st->eng, host_dst, r.dst_pitch, device_src, r.src_pitch, bytes, r.height, &st->dep, st->done);
        return dma::copy_to_host(st->eng, host_dst, device_src, bytes, &st->dep, st->done);
    }
    if (int rc = ensure_fetch_stream(ss)) return rc;
    PMT_HIP_CHECK(hipEventRecord(order_event, after));
    PMT_HIP_CHECK(hipStreamWaitEvent(ss->fetch, order_event, 0));
    // a <= 16-VGPR copy kernel that is co-resident with the contraction (deliver.hip) when the destination is page-locked, 8-byte words
    // and below 16 GiB; the runtime's copy otherwise
    const size_t total = r.height ? r.height * r.dst_pitch : bytes;
    const bool words = bytes % 8 == 0 && (!r.height || (r.dst_pitch % 8 == 0 && r.src_pitch % 8 == 0));
    void *dst_dev = (words && total / 8 < (size_t)1 << 31) ? host_device_pointer(host_dst) : nullptr;
    if (dst_dev && r.height) { if (int rc = launch_to_host_2d(device_src, r.src_pitch, dst_dev, r.dst_pitch, bytes, r.height, ss->fetch)) return rc; }
    else if (dst_dev) { if (int rc = launch_to_host(device_src, dst_dev, bytes, ss->fetch)) return rc; }
    else if (r.height) PMT_HIP_CHECK(hipMemcpy2DAsync(host_dst, r.dst_pitch, device_src, r.src_pitch, bytes, r.height, hipMemcpyDeviceToHost, ss->fetch));
    else PMT_HIP_CHECK(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ss->fetch));
    PMT_HIP_CHECK(hipEventRecord(ss->fetch_done, ss->fetch));
    ss->fetch_pending = true;
    return PMT_OK;
}

#ifdef PMT_TUNING
static double g_replay_t0 = 0;
static double host_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
// debugging aid (PMT_DMA_DEBUG=2): poll every pending completion signal and the calling stream, print when each changes
static void trace_dma_pending(SideStream *ss, hipStream_t s) {
    std::vector<long> last(ss->dma_pending.size(), -100);
    bool stream_done = false;
    const double t0 = host_us();
    for (;;) {
        bool all = true;
        for (size_t i = 0; i < ss->dma_pending.size(); ++i) {
            const long v = (long)*ss->dma_pending[i].second.value;
            if (v != last[i]) { fprintf(stderr, "[trace +%.0f us] transfer set %zu: %ld left\n", host_us() - g_replay_t0, i, v); last[i] = v; }
            if (v > 0) all = false;
        }
        if (!stream_done && hipStreamQuery(s) == hipSuccess) { stream_done = true; fprintf(stderr, "[trace +%.0f us] the plan's stream is idle\n", host_us() - g_replay_t0); }
        if ((all && stream_done) || host_us() - t0 > 1e6) break;
    }
}
#endif

static int wait_dma_pending(SideStream *ss) {
    int rc = PMT_OK;
    for (auto &p : ss->dma_pending) { const int r = dma::wait(p.first, p.second, 10.0); if (r && !rc) rc = r; }
    ss->dma_pending.clear();
    ss->keepalive.clear();
    return rc;
}

// `s` waits until the copies enqueued on its fetch stream so far have read their device buffers (start of the next re-evaluation)
int fetch_fence(hipStream_t s) {
    std::unique_lock<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it == g_side.end()) return PMT_OK;
    SideStream *ss = &it->second;
    lock.unlock();
    // copy-engine transfers are not stream work: the HOST waits for them (a no-op when the caller has synchronised, as solve! does)
    if (int rc = wait_dma_pending(ss)) return rc;
    if (!ss->fetch_pending) return PMT_OK;
    PMT_HIP_CHECK(hipStreamWaitEvent(s, ss->fetch_done, 0));
    ss->fetch_pending = false;
    return PMT_OK;
}

// a plan's replay brackets its tape with these: transfers that should queue up behind the tape's own are submitted by replay_end
void replay_begin(hipStream_t s) {
#ifdef PMT_TUNING
    g_replay_t0 = host_us();
#endif
    std::lock_guard<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it != g_side.end()) it->second.in_replay = true;
}
int replay_end(hipStream_t s) {
    std::unique_lock<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it == g_side.end()) return PMT_OK;
    SideStream *ss = &it->second;
    lock.unlock();
    ss->in_replay = false;
    int rc = PMT_OK;
    for (auto &f : ss->deferred) { const int r = f(); if (r && !rc) rc = r; }
    ss->deferred.clear();
    return rc;
}

// host: block until every copy enqueued on the fetch stream of `s` has landed
int fetch_synchronize(hipStream_t s) {
    std::unique_lock<std::mutex> lock(g_side_mu);
    auto it = g_side.find(s);
    if (it == g_side.end()) return PMT_OK;
    SideStream *ss = &it->second;
    hipStream_t f = ss->fetch;
    void *counters = ss->counters;
    lock.unlock();
#ifdef PMT_TUNING
    { const char *e = getenv("PMT_DMA_DEBUG"); if (e && e[0] == '2') trace_dma_pending(ss, s); }
#endif
    int rc = wait_dma_pending(ss);
    if (!rc && f) PMT_HIP_CHECK(hipStreamSynchronize(f));
    // the kernels' error word (page-locked, written with system-scope stores before the data the transfers above carried)
    const int err = ss->err_host ? __atomic_exchange_n(ss->err_host, 0, __ATOMIC_ACQ_REL) : 0;
    if (err == ERR_COURIER) {
        // the courier left without re-arming: flags back to "in the making", completion count to zero
        static const long long armed[MAXGROUPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        PMT_HIP_CHECK(hipMemset(counters, 0, COUNTER_BYTES));
        PMT_HIP_CHECK(hipMemcpy(static_cast<char *>(counters) + FLAGS_OFFSET, armed, sizeof armed, hipMemcpyHostToDevice));
        return fail(PMT_HIP_ERROR, "host delivery: the courier saw no progress of the contraction for 2 s and gave up; the host arrays are incomplete");
    }
    if (err == ERR_PAIR_FOLD)
        return fail(PMT_HIP_ERROR, "host delivery: a split tile of the contraction never received its first half (pair fold); the tile was "
                                   "written as NaN and the delivered quadratic coefficients are invalid");
    if (err) return fail(PMT_HIP_ERROR, "host delivery: unknown device error " + std::to_string(err));
    return rc;
}

hipStream_t side_stream_of(hipStream_t s) {
    SideStream *ss = side_stream(s);
    return ss ? ss->stream : nullptr;
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_quad_gram_constant_order(int64_t rows, int64_t cols, int *order, int *groups, int *stage_rows) {
    PMT_REQUIRE(rows >= 0 && cols >= 0 && order, PMT_INVALID_ARGUMENT, "quad_gram_constant_order: bad argument");
    if (cols > 0 && gram_mid_applies(rows, cols)) {          // the fix-up launch's last workgroup: 512 strided chains, wave trees, waves in order
        *order = 5;
        if (groups) *groups = 1;
        if (stage_rows) *stage_rows = 512;
        return PMT_OK;
    }
    const bool tall = cols > 0 && (gram_tall_applies(rows, cols) || gram_tall_diag_applies(rows, cols));
    *order = tall ? (gram_tall_run_lanes(rows, cols) == 4 ? 4 : gram_tall_run_lanes(rows, cols) == 16 ? 3 : 2) : (constant_chained(rows, cols) ? 1 : 0);
    if (groups) *groups = tall ? gram_tall_groups(rows, cols) : (*order == 1 ? 2048 : 1);
    if (stage_rows) *stage_rows = tall ? gram_tall_stage_rows(rows, cols) : 0;
    return PMT_OK;
}

extern "C" size_t pmt_quad_gram_workspace_bytes(int64_t rows, int64_t cols) {
    // behind the contraction's partial tiles: chunk sums of q (tall matrices) and the chains of the constant (long vectors); the fused
    // tall form (gram_tall.hip) keeps its per-chunk partials in the same buffer
    return std::max(std::max(gram_tall_workspace_bytes(rows, cols), gram_mid_applies(rows, cols) ? gram_mid_workspace_bytes(rows, cols) : (size_t)0),
                    gram_sk_workspace_bytes(rows, cols) + sizeof(double) * ((size_t)linear_splits(rows, cols) * (size_t)std::max<int64_t>(cols, 0) +
                                                                            blocked_dot_scratch_doubles()));
}

// the signals of one recorded delivery: a dependency signal per band group (a one-thread kernel behind the stage that completes the group
// stores 0 into its value when the group is in memory) and one completion signal that counts the groups' transfers down
struct DeliverSignals {
    dma::Engine *eng = nullptr;
    dma::Signal dep[MAXGROUPS], done;
    int n = 0;
    bool tried = false, pending = false;
    ~DeliverSignals() {
        if (!eng) return;
        for (int i = 0; i < n; ++i) dma::signal_destroy(eng, dep[i]);
        dma::signal_destroy(eng, done);
    }
};

// Host delivery of the CSC values (pmt_quad_gram_csc_deliver_f64): the contraction runs as a sequence of STAGES, each a launch over a range
// of the tile sequence (column-band-major, sk_colseq_unrank) followed by its fix-up; a stage that completes column bands ends with a one-thread
// kernel that releases the copy engine's transfer of those bands.
struct DeliverPlan {
    double *host = nullptr;                 // page-locked destination, same layout as out_csc
    double *host_dev = nullptr;             // ... and its device-visible address
    int nstages = 0;
    int64_t seq_begin[MAXGROUPS], seq_count[MAXGROUPS];
    int group_of[MAXGROUPS];                // index of the band group the stage completes, or -1
    int ngroups = 0;
    int64_t gbeg[MAXGROUPS], gend[MAXGROUPS];      // the groups' ranges of the delivered array (doubles)
};

// Stage size.  With the persistent grid all workgroups finish a round of whole tiles together, so one launch of everything delivers in two
// bursts (at n = r = 4096: nothing for 0.6 ms, half of P then, the rest at the end: 1.95 ms per solve).  Stages of HALF a grid's worth of
// tiles — every tile split in two along the contraction, stream-K over all workgroups, partial sums added by the fix-up launch — complete
// a quarter of P every 0.3 ms, which is also what PCIe takes to ship it: the copy engine stays busy from the first stage on.  The split
// costs ~15 % of contraction time (partial tiles through the workspace, 3 launches per stage; profiles/r03_host_delivery.txt) and changes
// the summation order of a split tile (two half sums added) — within the stated tolerance, deterministic, and the delivered host array is
// the device array of the same run bit for bit.  `nstages_hint` (the entry point's ngroups) > 0 overrides the number of stages.
// quad: the delivered array is out_quad (24-byte terms, ROW-major upper triangle): tile order 0 (super-rows of four tile rows, sk_seq_unrank), the
// groups are row bands and the offsets count doubles (three per term); else out_csc (column bands, super-columns of `order_w`).
static DeliverPlan deliver_plan(int64_t rows, int64_t cols, int nstages_hint, int order_w, double *host, bool quad) {
    DeliverPlan d;
    d.host = host;
    const int nt = (int)cdiv(cols, GT);
    const int64_t T = (int64_t)nt * (nt + 1) / 2;
    const int64_t nchunk = std::max<int64_t>(1, cdiv(rows, 256));
    const int64_t G = std::min<int64_t>(T * nchunk, 256);                  // as launch_gram_sk (gram_sk.hip)
    int64_t per = nchunk >= 2 ? std::max<int64_t>(1, G / 2) : G;           // tiles per stage: half a grid's worth (whole tiles if there is nothing to split)
    if (nstages_hint > 0) per = std::max<int64_t>(1, cdiv(T, std::min(nstages_hint, MAXGROUPS)));
    if (cdiv(T, per) > MAXGROUPS) per = cdiv(T, MAXGROUPS);
    // position in the walk after which each band is complete, and the band's range of the delivered array
    std::vector<int64_t> seq_end_of_band((size_t)nt), bandbeg((size_t)nt), bandend((size_t)nt);
    int64_t seq_end = 0;
    if (quad) {
        // row bands, ascending (row band jb holds nt - jb tiles: the short ones come last by themselves)
        for (int j0 = 0; j0 < nt; j0 += 4) {
            const int h = std::min(4, nt - j0), W = nt - j0;
            seq_end += (int64_t)h * (h + 1) / 2 + (int64_t)(W - h) * h;    // (sk_seq_unrank)
            for (int jb = j0; jb < j0 + h; ++jb) seq_end_of_band[(size_t)jb] = seq_end;
        }
        for (int jb = 0; jb < nt; ++jb) {
            const int64_t J0 = (int64_t)jb * GT, J = std::min<int64_t>(cols, (int64_t)(jb + 1) * GT);
            bandbeg[(size_t)jb] = 3 * (J0 * cols - J0 * (J0 - 1) / 2);      // rows 0 .. J-1 of the row-major upper triangle, 3 doubles per term
            bandend[(size_t)jb] = 3 * (J * cols - J * (J - 1) / 2);
        }
    } else {
        // column bands, DESCENDING: the walk starts at the end of the column-band-major sequence (sk_colseq_unrank, launch_gram_sk order_w < 0)
        std::vector<int64_t> start_of_super((size_t)nt);
        int64_t pos = 0;
        for (int c0 = 0; c0 < nt; c0 += order_w) {
            const int h = std::min(order_w, nt - c0);
            for (int kb = c0; kb < c0 + h; ++kb) start_of_super[(size_t)kb] = pos;
            pos += (int64_t)c0 * h + (int64_t)h * (h + 1) / 2;
        }
        for (int kb = 0; kb < nt; ++kb) {
            seq_end_of_band[(size_t)kb] = T - start_of_super[(size_t)kb];     // the walk has passed the band's whole super-column
            const int64_t c0 = (int64_t)kb * GT, cend = std::min<int64_t>(cols, (int64_t)(kb + 1) * GT);
            bandbeg[(size_t)kb] = c0 * (c0 + 1) / 2;
            bandend[(size_t)kb] = cend * (cend + 1) / 2;
        }
    }
    // bands in completion order: 0, 1, .. (quad) or nt-1, nt-2, .. (CSC)
    auto band_at = [&](int i) { return quad ? i : nt - 1 - i; };
    int done_bands = 0;
    for (int64_t t0 = 0; t0 < T; t0 += per) {
        const int st = d.nstages++;
        d.seq_begin[st] = t0;
        d.seq_count[st] = std::min<int64_t>(per, T - t0);
        int nb = done_bands;
        while (nb < nt && seq_end_of_band[(size_t)band_at(nb)] <= t0 + d.seq_count[st]) ++nb;
        d.group_of[st] = -1;
        if (nb > done_bands) {
            d.group_of[st] = d.ngroups;
            const int first = band_at(done_bands), last = band_at(nb - 1);
            d.gbeg[d.ngroups] = bandbeg[(size_t)std::min(first, last)];
            d.gend[d.ngroups] = bandend[(size_t)std::max(first, last)];
            ++d.ngroups;
            done_bands = nb;
        }
    }
    return d;
}

// the whole node: contraction on the main stream, q = 2A'c and c'c on a side stream.  out_quad (term structs) and out_csc (solver
// values, alpha-scaled) are independent optional outputs of the same contraction.  host_csc != null: out_csc is also DELIVERED to that
// page-locked host buffer, band group by band group, on the stream's fetch stream while the contraction is still running.
static int gram_node(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                     int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                     double *out_const, void *workspace, double *host_csc, int ngroups, void *stream, pmt_quadratic_term *host_quad = nullptr) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "quad_gram: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "quad_gram: lda < rows");
    PMT_REQUIRE(sign >= -1 && sign <= 1, PMT_INVALID_ARGUMENT, "quad_gram: sign must be -1, 0 or +1");
    PMT_REQUIRE(out_const, PMT_INVALID_ARGUMENT, "quad_gram: null out_const");
    PMT_REQUIRE(sign == 0 || b || rows == 0, PMT_INVALID_ARGUMENT, "quad_gram: sign != 0 needs b");
    if (cols > 0) PMT_REQUIRE(xvar && (out_quad || out_csc) && out_lin && (A || rows == 0), PMT_INVALID_ARGUMENT, "quad_gram: null pointer");
    PMT_REQUIRE(cols < (int64_t)GT * 32000, PMT_DIMENSION_MISMATCH, "quad_gram: too many columns");
    if (int rc = check_strictly_increasing(xvar, cols, stream)) return rc;
    DeliverPlan dplan;
    std::shared_ptr<DeliverSignals> sig = std::make_shared<DeliverSignals>();      // lives as long as the recorded call
    PMT_REQUIRE(!(host_csc && host_quad), PMT_INVALID_ARGUMENT, "quad_gram: one delivered array per call");
    // the array a delivery ships, as doubles: out_csc, or out_quad (three doubles per term)
    double *deliver_host = host_quad ? reinterpret_cast<double *>(host_quad) : host_csc;
    const double *deliver_src = host_quad ? reinterpret_cast<const double *>(out_quad) : out_csc;
#ifdef PMT_TUNING
    static const int deliver_w = [] { const char *e = getenv("PMT_DELIVER_ORDER_W"); return e ? atoi(e) : DELIVER_ORDER_W; }();
#else
    constexpr int deliver_w = DELIVER_ORDER_W;
#endif
    const int deliver_order = host_quad ? 0 : -deliver_w;
    // shapes of up to 2048 columns take the fused tall forms (gram_tall.hip) whatever the outputs: one 128-column tile — triangle, q and c'c in
    // ONE pass over A, no side stream, no separate reductions —, or several — the diagonal tiles, q and c'c in one fused pass, then the strictly
    // upper tiles in ONE ranged launch of the stream-K kernel (SKArgs::strict: its tile sequence leaves the diagonal out; the partials of both
    // forms share the workspace in stream order).  A host delivery of such a node is ONE transfer of the whole array behind it: the staged
    // contraction pays for hundreds of megabytes (config 2), not for the <= 50 MB of these shapes.
    // (2049 .. 4096 columns: the one-launch form unless a staged delivery is asked for — gram_mid_big)
    const bool mid_big = cols > 0 && workspace && !deliver_host && gram_mid_big(rows, cols);
    const bool tall_form = cols > 0 && workspace && (gram_tall_applies(rows, cols) || gram_tall_diag_applies(rows, cols) || mid_big);
    if (deliver_host && cols > 0) {
        if (tall_form) {
            dplan.host = deliver_host;
            dplan.nstages = 0; dplan.ngroups = 1; dplan.gbeg[0] = 0;
            dplan.gend[0] = (host_quad ? 3 : 1) * (cols * (cols + 1) / 2);
        } else dplan = deliver_plan(rows, cols, ngroups, deliver_w, deliver_host, host_quad != nullptr);
        dplan.host_dev = static_cast<double *>(host_device_pointer(deliver_host));
        PMT_REQUIRE(dplan.host_dev, PMT_INVALID_ARGUMENT, "quad_gram_csc_deliver: host_P_values must be page-locked host memory (pmt_host_alloc)");
        mark_no_graph(stream);
    }
    Launch node = [=](hipStream_t s) {
        // fork: the two small reductions of this node (q = 2 A'c, HBM-bound; c'c, a serial chain) run on a side stream while
        // the MFMA-bound contraction owns the main stream; join before returning control of `s`.  Legal under stream capture.
        SideStream *side = side_stream(s);
        const bool deliver = dplan.host != nullptr;
        bool use_engine = false;
        if (deliver) {
            PMT_REQUIRE(side && side->counters, PMT_STATE_ERROR, "quad_gram_csc_deliver: no auxiliary streams for this stream");
            // Preferred: one copy-engine transfer per band group, each started by the signal set behind the group's stage (hsadma.hip)
            const int mode = dma::delivery_mode();
            // the previous delivery of this entry must have read its array before this contraction overwrites it (whichever way it went)
            if (sig->pending) { if (int rc = dma::wait(sig->eng, sig->done, 10.0)) return rc; sig->pending = false; }
            // an immediate (unrecorded) call owns fresh signals: what earlier calls on this stream handed to the engine is awaited here
            if (!side->in_replay) { if (int rc = wait_dma_pending(side)) return rc; }
            if (mode != 2 && !sig->tried) {
                sig->tried = true;
                dma::Engine *eng = dma::get(side->device);
                if (eng) {
                    bool ok = dma::signal_create(eng, 0, &sig->done) == PMT_OK;
                    for (int i = 0; ok && i < dplan.ngroups; ++i) { ok = dma::signal_create(eng, 1, &sig->dep[i]) == PMT_OK; if (ok) sig->n = i + 1; }
                    if (ok) sig->eng = eng;
                }
            }
            use_engine = mode != 2 && sig->eng != nullptr;
            PMT_REQUIRE(mode != 1 || use_engine, PMT_STATE_ERROR, "host delivery: the copy engine was demanded (pmt_set_host_delivery(1)) but is not available");
            if (use_engine) {
                // (a previous delivery that went through the courier — the mode was switched in between — is ordered by its event)
                if (side->fetch_pending) { PMT_HIP_CHECK(hipStreamWaitEvent(s, side->fetch_done, 0)); side->fetch_pending = false; }
                for (int i = 0; i < dplan.ngroups; ++i) dma::signal_set(sig->eng, sig->dep[i], 1);
                dma::signal_set(sig->eng, sig->done, dplan.ngroups);
            } else {
                if (int rc = ensure_fetch_stream(side)) return rc;
                if (side->fetch_pending) { PMT_HIP_CHECK(hipStreamWaitEvent(s, side->fetch_done, 0)); side->fetch_pending = false; }
            }
        }
        hipStream_t s2 = s;
        if (side && !tall_form) {
            PMT_HIP_CHECK(hipEventRecord(side->fork, s));
            PMT_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
            s2 = side->stream;
        }
        int rc = PMT_OK;
        double *scratch = workspace ? reinterpret_cast<double *>(static_cast<char *>(workspace) + gram_sk_workspace_bytes(rows, cols)) : nullptr;
        const int nsplit = (scratch && b && sign) ? linear_splits(rows, cols) : 1;
        if (tall_form || PMT_GRAM_ABL_NO_LINEAR) {
        } else if (cols > 0 && nsplit > 1) {
            const int64_t chunk = 64 * cdiv(cdiv(rows, nsplit), 64);
            PMT_LAUNCH(gram_linear_split_kernel, dim3((unsigned)cdiv(cols, 4), (unsigned)nsplit), dim3(256), 0, s2, A, lda, rows, cols, b, sign, chunk, scratch);
            PMT_LAUNCH(gram_linear_finish_kernel, dim3((unsigned)cdiv(cols, 256)), dim3(256), 0, s2, scratch, nsplit, cols, xvar, moi, varmap, out_lin);
            rc = check_launch("gram_linear_split_kernel");
        } else if (cols > 0) {
            PMT_LAUNCH(gram_linear_kernel, dim3((unsigned)cdiv(cols, 4)), dim3(256), 0, s2, A, lda, rows, cols, xvar, b, sign, moi, varmap, out_lin);
            rc = check_launch("gram_linear_kernel");
        }
        // c'c: a serial chain of `rows` additions on ONE wave (bit for bit the reference's left-to-right sum) — ~50 us alone, ~0.3 ms beside
        // the contraction.  Inside a plan's replay it is queued at the END of the replay, i.e. behind the tape's side-lane entries on the
        // side stream (the MOI copies of the constraints, the hand-off gathers and their fetches), which used to wait for it.
        double *chains = scratch ? scratch + (size_t)linear_splits(rows, cols) * (size_t)cols : nullptr;
        auto const_part = [=]() -> int {
            int rc2 = PMT_OK;
            if (gram_mid_big(rows, cols)) rc2 = launch_gram_mid_constant(b, sign, rows, out_const, s2);      // (a staged delivery: the one-launch form's order)
            else if (b && sign && rows > 0) rc2 = launch_blocked_dot(b, sign, b, sign, rows, chains, out_const, s2, constant_chained(rows, cols) ? 1 : 0);
            else if (hipMemsetAsync(out_const, 0, sizeof(double), s2) != hipSuccess) rc2 = fail(PMT_HIP_ERROR, "hipMemsetAsync(out_const)");
            if (side) {
                PMT_HIP_CHECK(hipEventRecord(side->join2, side->stream));
                PMT_HIP_CHECK(hipStreamWaitEvent(s, side->join2, 0));
            }
            return rc2;
        };
        const bool defer_const = side && side->in_replay && !tall_form;
        if (!rc && defer_const) side->deferred.push_back(const_part);
        if (side && !tall_form) PMT_HIP_CHECK(hipEventRecord(side->join, side->stream));          // the affine part: `s` joins it behind the contraction's launch
        if (!rc && cols > 0) {
            if (tall_form && (mid_big || (gram_mid_applies(rows, cols) && !gram_mid_big(rows, cols)))) {
                // WIDE shapes of up to 2048 columns (gram_mid_applies): the whole node — every tile, q and c'c — in ONE launch on 64 x 64 tiles
                // (gram_mid.hip); the per-tile arrival counts are this calling stream's
                PMT_REQUIRE(side && side->counters && (size_t)gram_mid_counters(cols) * sizeof(unsigned) <= MID_COUNTER_BYTES, PMT_STATE_ERROR,
                            "quad_gram: no auxiliary state for this stream");
                rc = launch_gram_mid(A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, out_csc, alpha, out_lin, out_const, workspace,
                                     reinterpret_cast<unsigned *>(static_cast<char *>(side->counters) + MID_OFFSET), s);
                if (!rc && side && side->in_replay) {          // side-lane entries behind this node may read its affine part (see below)
                    PMT_HIP_CHECK(hipEventRecord(side->fork, s));
                    PMT_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
                }
                if (!rc && deliver) {
                    char *cb = static_cast<char *>(side->counters);
                    dma::Signal word = use_engine ? sig->dep[0] : dma::Signal{0, reinterpret_cast<int64_t *>(cb + FLAGS_OFFSET)};
                    rc = dma::launch_signal_store(word, s);
                }
            } else
            if (tall_form) {
                rc = launch_gram_tall(A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, out_csc, alpha, out_lin, out_const, workspace, s);
                // a plan's side-lane entries recorded behind this node may read its AFFINE part (the hand-off's q gather; plan.hip `replay`): with
                // the stream-K form they queue behind gram_linear on the side stream — here the side stream is told to wait for the fix-up
                // that has just written q and the constant (the strictly upper tiles below then run beside those entries)
                if (!rc && side && side->in_replay) {
                    PMT_HIP_CHECK(hipEventRecord(side->fork, s));
                    PMT_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
                }
                const int64_t nt = cdiv(cols, GT);
                if (!rc && nt > 1)
                    rc = launch_gram_sk(A, lda, rows, cols, xvar, varmap, moi, out_quad, out_csc, alpha, workspace, 0, 0, nt * (nt - 1) / 2, nullptr, 0, nullptr, s, 1);
                if (!rc && deliver) {                 // the whole array is in memory behind the node's last kernel: release its one transfer
                    char *cb = static_cast<char *>(side->counters);
                    dma::Signal word = use_engine ? sig->dep[0] : dma::Signal{0, reinterpret_cast<int64_t *>(cb + FLAGS_OFFSET)};
                    rc = dma::launch_signal_store(word, s);
                }
            } else if (!deliver) {
                rc = launch_gram_sk(A, lda, rows, cols, xvar, varmap, moi, out_quad, out_csc, alpha, workspace, 0, 0, -1, nullptr, 0, nullptr, s);
            } else {
                // stage by stage; behind a stage that completes column bands, one thread stores 0 into the word the transfer of those bands waits
                // for: the value of its dependency signal (copy engine, hsadma.hip) or the courier's flag (deliver.hip)
                char *cb = static_cast<char *>(side->counters);
                // flags of the pair fold (gram_sk.hip): the tail of the partial-tile workspace, beyond the slots a grid of 256 uses; cleared
                // per delivery, stage st writes / waits for the value st + 1
                unsigned *pair_flags = workspace ? reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + gram_sk_workspace_bytes(rows, cols) - PAIR_FLAG_BYTES) : nullptr;
                if (pair_flags) PMT_HIP_CHECK(hipMemsetAsync(pair_flags, 0, PAIR_FLAG_BYTES, s));
                for (int st = 0; !rc && st < dplan.nstages; ++st) {
                    rc = launch_gram_sk(A, lda, rows, cols, xvar, varmap, moi, out_quad, out_csc, alpha, workspace, deliver_order, dplan.seq_begin[st],
                                        dplan.seq_count[st], pair_flags, (unsigned)(st + 1), side->err_dev, s);
                    const int grp = dplan.group_of[st];
                    if (!rc && grp >= 0) {
                        dma::Signal word = use_engine ? sig->dep[grp] : dma::Signal{0, reinterpret_cast<int64_t *>(cb + FLAGS_OFFSET) + grp};
                        rc = dma::launch_signal_store(word, s);
                    }
                }
            }
            if (!rc && deliver) {
                if (use_engine) {
                    // The engine works through its queue in submission order.  Inside a plan's replay the groups' transfers are therefore
                    // submitted at the END of the replay, behind the recorded fetches of the tape (q, A's values, bounds: ready within the
                    // first tenth of the contraction) — submitted here they would hold those back until the last band group has left.
                    auto submit = [=]() -> int {
                        for (int i = 0; i < dplan.ngroups; ++i)
                            if (int rc2 = dma::copy_to_host(sig->eng, dplan.host + dplan.gbeg[i], deliver_src + dplan.gbeg[i],
                                                            sizeof(double) * (size_t)(dplan.gend[i] - dplan.gbeg[i]), &sig->dep[i], sig->done, 1)) return rc2;
                        return PMT_OK;
                    };
                    sig->pending = true;
                    side->dma_pending.emplace_back(sig->eng, sig->done);
                    side->keepalive.push_back(sig);
                    if (side->in_replay) side->deferred.push_back(submit);
                    else if (int rc2 = submit()) return rc2;
                } else {
                    // fallback, fetch stream: ONE courier launch, queued now that the contraction's workgroups are on their way; it polls the
                    // band groups' flags and stores each finished group straight into the host array (deliver.hip)
                    char *cb = static_cast<char *>(side->counters);
                    rc = launch_courier(deliver_src, dplan.host_dev, reinterpret_cast<long long *>(cb + FLAGS_OFFSET), reinterpret_cast<unsigned *>(cb + DONE_OFFSET),
                                        side->err_dev, dplan.ngroups, dplan.gbeg, dplan.gend, side->fetch);
                    if (rc) return rc;
                    PMT_HIP_CHECK(hipEventRecord(side->fetch_done, side->fetch));
                    side->fetch_pending = true;
                }
            }
        }
        if (tall_form) return rc;
        if (side) PMT_HIP_CHECK(hipStreamWaitEvent(s, side->join, 0));
        if (!rc && !defer_const) rc = const_part();          // (behind the contraction's launch: its workgroups are placed first)
        return rc;
    };
    // tiny shapes (README Example 1 with the canonical objective: 8 x 8): also a small-plan node — row-order sums by the interpreter kernel
    // instead of four launches and a side-stream fork
    if (!deliver_host && !out_csc && gram_tiny(rows, cols)) {
        SmallNode nd;
        nd.op = SOP_GRAM; nd.sign = (b && sign) ? sign : 0; nd.moi = moi; nd.d[0] = lda; nd.d[1] = rows; nd.d[2] = cols;
        nd.in[0] = A; nd.in[1] = xvar; nd.in[2] = b; nd.in[3] = varmap; nd.out[0] = out_quad; nd.out[1] = out_lin; nd.out[2] = out_const;
        nd.work = rows * cols * (cols + 1) / 2 + 64 * rows;
        // (by itself — an immediate call, or a run of one — the node is ONE launch of the interpreter's body: the same bits as inside a run)
        return dispatch(stream, [=](hipStream_t s) { return launch_small_one(nd, s); }, nd);
    }
    return dispatch(stream, node);
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

extern "C" int pmt_quad_gram_deliver_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                         int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, pmt_quadratic_term *host_quad, int nstages,
                                         pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_quad && host_quad, PMT_INVALID_ARGUMENT, "quad_gram_deliver: null out_quad / host_quad");
    PMT_REQUIRE(nstages >= 0 && nstages <= MAXGROUPS, PMT_INVALID_ARGUMENT, "quad_gram_deliver: nstages must be 0 (default) .. 16");
    return gram_node(A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, nullptr, 1.0, out_lin, out_const, workspace, nullptr, nstages, stream, host_quad);
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
