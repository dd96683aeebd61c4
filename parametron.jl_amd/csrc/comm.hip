// Multi-GPU exchange for the batched configuration (BASELINE config 4): every rank rebuilds the coefficient slabs of
// its own contiguous range of instances and every rank ends up with every slab.  In the reference a batch is many independent Models
// (src/model.jl:1-22) — there is no exchange code to restate; this is the one collective of the whole path.
//
// RCCL is driven from HERE, behind the C ABI (a Julia host reaches it with two ccalls; torch.distributed is only the launcher that
// carries the 128-byte unique id to the other ranks).  librccl is dlopen'ed on first use, not linked: the library stays loadable on a box
// without RCCL, and a process that already holds an RCCL (PyTorch's) shares that one instead of loading a second copy.
//
// Schedule (xGMI is point-to-point, 7 links per GPU, ~153 GB/s each): a ring all-gather of the 85.6 MB slab block is per-link bound at
// 7 x 85.6 MB / 153 GB/s = 3.9 ms; the DIRECT schedule — every rank sends its block to each of the other 7 over that peer's own link —
// is 85.6 MB / 153 GB/s = 0.56 ms.  pmt_batch_allgather_f64 therefore issues one grouped ncclSend/ncclRecv pair per peer (RCCL maps
// them to the direct links), and pmt_batch_step_f64 overlaps it with the computation: the local instances are processed in chunks, and
// chunk c is on the wire (communication stream) while chunk c + 1 is being computed (compute stream).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace pmt {

int launch_batch_small(const double *A, int64_t lda, int64_t rows, int64_t cols, int64_t strideA, const double *b, int64_t strideb, int sign,
                       int64_t B, double *out, int64_t out_stride, const double *Cm, int64_t m, const double *d, int sign_d, hipStream_t s);

namespace {

// the slice of the NCCL/RCCL API that is used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclDouble = 8, ncclSuccess = 0)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId *);
typedef int (*CommInitRank_t)(void **, int, UniqueId, int);
typedef int (*CommDestroy_t)(void *);
typedef int (*GroupStart_t)();
typedef int (*GroupEnd_t)();
typedef int (*Send_t)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*Recv_t)(void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*GetErrorString_t)(int);
constexpr int kNcclDouble = 8;

struct Rccl {
    void *handle = nullptr;
    GetUniqueId_t get_unique_id = nullptr;
    CommInitRank_t comm_init_rank = nullptr;
    CommDestroy_t comm_destroy = nullptr;
    GroupStart_t group_start = nullptr;
    GroupEnd_t group_end = nullptr;
    Send_t send = nullptr;
    Recv_t recv = nullptr;
    GetErrorString_t error_string = nullptr;
    std::string why;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + n; return p; };
        r.get_unique_id = (GetUniqueId_t)sym("ncclGetUniqueId");
        r.comm_init_rank = (CommInitRank_t)sym("ncclCommInitRank");
        r.comm_destroy = (CommDestroy_t)sym("ncclCommDestroy");
        r.group_start = (GroupStart_t)sym("ncclGroupStart");
        r.group_end = (GroupEnd_t)sym("ncclGroupEnd");
        r.send = (Send_t)sym("ncclSend");
        r.recv = (Recv_t)sym("ncclRecv");
        r.error_string = (GetErrorString_t)sym("ncclGetErrorString");
    });
    return &r;
}

int rccl_fail(const char *what, int code) {
    Rccl *r = rccl();
    return fail(PMT_HIP_ERROR, std::string(what) + ": " + (r->error_string ? r->error_string(code) : "RCCL error ") + " (" + std::to_string(code) + ")");
}

#define PMT_RCCL_READY()                                                                         \
    Rccl *R = rccl();                                                                            \
    if (!R->handle || !R->why.empty()) return fail(PMT_STATE_ERROR, "RCCL is not available: " + R->why)
#define PMT_RCCL_CHECK(expr)                                     \
    do {                                                         \
        int _c = (expr);                                         \
        if (_c != 0) return rccl_fail(#expr, _c);                \
    } while (0)

}  // namespace

struct Comm {
    void *nccl = nullptr;                    // null: a single rank without RCCL (nothing to exchange, nothing loaded)
    int rank = 0, nranks = 1, device = 0;
    int64_t rccl_calls = 0;                  // ncclSend + ncclRecv issued so far (pmt_comm_rccl_calls)
    hipStream_t comm_stream = nullptr;       // the exchange runs here, beside the computation on the caller's stream
    hipEvent_t computed = nullptr, gathered = nullptr;
};

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_comm_unique_id(void *out_id_128_bytes) {
    PMT_REQUIRE(out_id_128_bytes, PMT_INVALID_ARGUMENT, "comm_unique_id: null pointer");
    PMT_RCCL_READY();
    PMT_RCCL_CHECK(R->get_unique_id(reinterpret_cast<UniqueId *>(out_id_128_bytes)));
    return PMT_OK;
}

extern "C" int pmt_comm_destroy(void *comm);

extern "C" int pmt_comm_init_rank(int nranks, int rank, const void *unique_id_128_bytes, int device, void **out_comm) {
    PMT_REQUIRE(out_comm && nranks >= 1 && rank >= 0 && rank < nranks, PMT_INVALID_ARGUMENT, "comm_init_rank: bad argument");
    PMT_REQUIRE(nranks == 1 || unique_id_128_bytes, PMT_INVALID_ARGUMENT, "comm_init_rank: null unique id");
    Rccl *R = nullptr;
    // A single rank needs no communicator (and no RCCL) — unless it brings a unique id: then it gets a real one-rank RCCL communicator and
    // its exchange is a grouped ncclSend/ncclRecv to itself, i.e. the code path of N ranks on one GPU (how the binding is tested on a
    // single-GPU box).  Checked BEFORE anything is allocated.
    const bool use_rccl = nranks > 1 || unique_id_128_bytes != nullptr;
    if (use_rccl) {
        R = rccl();
        if (!R->handle || !R->why.empty()) return fail(PMT_STATE_ERROR, "RCCL is not available: " + R->why);
    }
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->computed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->gathered, hipEventDisableTiming);
    // every failure from here on leaves through pmt_comm_destroy: the stream and the events go with the communicator
    if (e != hipSuccess) { (void)hipGetLastError(); (void)pmt_comm_destroy(c); return fail(PMT_HIP_ERROR, std::string("comm_init_rank: ") + hipGetErrorString(e)); }
    if (use_rccl) {
        UniqueId id;
        memcpy(&id, unique_id_128_bytes, sizeof id);
        int rc = R->comm_init_rank(&c->nccl, nranks, id, rank);
        if (rc != 0) { c->nccl = nullptr; (void)pmt_comm_destroy(c); return rccl_fail("ncclCommInitRank", rc); }
    }
    *out_comm = c;
    return PMT_OK;
}

extern "C" int64_t pmt_comm_rccl_calls(void *comm) { return comm ? reinterpret_cast<Comm *>(comm)->rccl_calls : 0; }

extern "C" int pmt_comm_destroy(void *comm) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return PMT_OK;
    (void)hipSetDevice(c->device);
    if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
    if (c->computed) (void)hipEventDestroy(c->computed);
    if (c->gathered) (void)hipEventDestroy(c->gathered);
    if (c->nccl) { Rccl *R = rccl(); if (R->comm_destroy) (void)R->comm_destroy(c->nccl); }
    delete c;
    return PMT_OK;
}

// ---- the chunk schedule (pure host arithmetic: the 2-rank gloo test on CPU drives the same functions) ----------------------------------
// rank g owns the instances [g * per_rank, (g + 1) * per_rank) of the batch; its chunk c is the local range [c * chunk, min(., per_rank));
// in the gathered buffer the slab of global instance i sits at i * stride doubles.
extern "C" int pmt_batch_shard(int64_t total, int nranks, int rank, int64_t *per_rank, int64_t *first) {
    PMT_REQUIRE(total >= 0 && nranks >= 1 && per_rank && first, PMT_INVALID_ARGUMENT, "batch_shard: bad argument");
    PMT_REQUIRE(rank >= 0 && rank < nranks, PMT_INVALID_ARGUMENT, "batch_shard: rank outside the communicator");
    PMT_REQUIRE(total % nranks == 0, PMT_DIMENSION_MISMATCH,
                "batch_shard: " + std::to_string(total) + " instances do not divide over " + std::to_string(nranks) + " ranks (the exchange needs equal shards)");
    *per_rank = total / nranks;
    *first = *per_rank * rank;
    return PMT_OK;
}

extern "C" int64_t pmt_batch_num_chunks(int64_t per_rank, int64_t chunk) {
    if (per_rank <= 0) return 0;
    if (chunk <= 0 || chunk > per_rank) chunk = per_rank;
    return (per_rank + chunk - 1) / chunk;
}

extern "C" int pmt_batch_chunk_range(int64_t per_rank, int64_t chunk, int64_t c, int64_t *lo, int64_t *hi) {
    PMT_REQUIRE(lo && hi, PMT_INVALID_ARGUMENT, "batch_chunk_range: null pointer");
    const int64_t nc = pmt_batch_num_chunks(per_rank, chunk);
    PMT_REQUIRE(c >= 0 && c < nc, PMT_INVALID_ARGUMENT, "batch_chunk_range: chunk index out of range");
    if (chunk <= 0 || chunk > per_rank) chunk = per_rank;
    *lo = c * chunk;
    *hi = std::min(per_rank, (c + 1) * chunk);
    return PMT_OK;
}

extern "C" int64_t pmt_batch_gathered_offset(int rank, int64_t per_rank, int64_t local_instance, int64_t stride) {
    return ((int64_t)rank * per_rank + local_instance) * stride;
}

// exchange of the local instances [lo, hi): `local` holds this rank's slabs (per_rank x stride doubles), `gathered` all of them
// (nranks * per_rank x stride).  Enqueued on `stream`; one grouped send/recv pair per peer (direct schedule).
static int exchange_range(Comm *c, const double *local, double *gathered, int64_t per_rank, int64_t stride, int64_t lo, int64_t hi, hipStream_t stream) {
    const size_t count = (size_t)((hi - lo) * stride);
    if (count == 0) return PMT_OK;
    double *mine = gathered + pmt_batch_gathered_offset(c->rank, per_rank, lo, stride);
    const double *src = local + lo * stride;
    if (c->nranks == 1 && c->nccl) {
        // one-rank RCCL communicator: this rank's own block travels through RCCL (send to self / receive from self in one group)
        PMT_RCCL_READY();
        PMT_RCCL_CHECK(R->group_start());
        PMT_RCCL_CHECK(R->send(src, count, kNcclDouble, 0, c->nccl, stream));
        PMT_RCCL_CHECK(R->recv(mine, count, kNcclDouble, 0, c->nccl, stream));
        PMT_RCCL_CHECK(R->group_end());
        c->rccl_calls += 2;
        return PMT_OK;
    }
    if (mine != src) PMT_HIP_CHECK(hipMemcpyAsync(mine, src, count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (c->nranks == 1) return PMT_OK;
    PMT_RCCL_READY();
    PMT_RCCL_CHECK(R->group_start());
    c->rccl_calls += 2 * (c->nranks - 1);
    for (int k = 1; k < c->nranks; ++k) {
        const int to = (c->rank + k) % c->nranks, from = (c->rank - k + c->nranks) % c->nranks;       // staggered: every link busy in every step
        PMT_RCCL_CHECK(R->send(src, count, kNcclDouble, to, c->nccl, stream));
        PMT_RCCL_CHECK(R->recv(gathered + pmt_batch_gathered_offset(from, per_rank, lo, stride), count, kNcclDouble, from, c->nccl, stream));
    }
    PMT_RCCL_CHECK(R->group_end());
    return PMT_OK;
}

extern "C" int pmt_batch_allgather_f64(void *comm, const double *local, double *gathered, int64_t per_rank, int64_t stride, int64_t chunk, void *stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    PMT_REQUIRE(c && per_rank >= 0 && stride >= 0, PMT_INVALID_ARGUMENT, "batch_allgather: bad argument");
    if (per_rank == 0 || stride == 0) return PMT_OK;
    PMT_REQUIRE(local && gathered, PMT_INVALID_ARGUMENT, "batch_allgather: null pointer");
    PMT_HIP_CHECK(hipSetDevice(c->device));
    const int64_t nc = pmt_batch_num_chunks(per_rank, chunk);
    for (int64_t k = 0; k < nc; ++k) {
        int64_t lo, hi;
        if (int rc = pmt_batch_chunk_range(per_rank, chunk, k, &lo, &hi)) return rc;
        if (int rc = exchange_range(c, local, gathered, per_rank, stride, lo, hi, reinterpret_cast<hipStream_t>(stream))) return rc;
    }
    return PMT_OK;
}

// One re-evaluation of this rank's share of the batch with the exchange overlapped: chunk k is computed on `stream`, then handed to the
// communication stream (event), which sends it while `stream` already computes chunk k + 1.  On return everything is ENQUEUED; `stream`
// waits for the last exchange, so work the caller enqueues on `stream` afterwards sees the complete `gathered` buffer.
extern "C" int pmt_batch_step_f64(void *comm, const double *A, const double *b, const double *Cm, const double *d, int64_t per_rank, int64_t n,
                                  int64_t r, int64_t m, int sign_b, int sign_d, double *local, double *gathered, int64_t stride, int64_t chunk,
                                  void *stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    PMT_REQUIRE(c && per_rank >= 0 && n >= 0 && r >= 0 && m >= 0, PMT_INVALID_ARGUMENT, "batch_step: bad argument");
    const int64_t L = pmt_batch_lsq_slab_doubles(n, m);
    PMT_REQUIRE(stride >= L, PMT_DIMENSION_MISMATCH, "batch_step: stride smaller than the slab");
    if (per_rank == 0) return PMT_OK;
    PMT_REQUIRE(local && gathered, PMT_INVALID_ARGUMENT, "batch_step: null pointer");
    PMT_REQUIRE(!is_recording_handle(stream), PMT_STATE_ERROR, "batch_step: needs a HIP stream, not a plan's recording handle (it orders two streams with events)");
    PMT_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t nc = pmt_batch_num_chunks(per_rank, chunk);
    // the previous step's exchange has been joined into `s` (below), so `local` / `gathered` are free to be rewritten in stream order
    for (int64_t k = 0; k < nc; ++k) {
        int64_t lo, hi;
        if (int rc = pmt_batch_chunk_range(per_rank, chunk, k, &lo, &hi)) return rc;
        int rc = pmt_batch_lsq_coeffs_f64(A + lo * r * n, b + lo * r, Cm ? Cm + lo * m * n : nullptr, d ? d + lo * m : nullptr, hi - lo, n, r, m, sign_b,
                                          sign_d, local + lo * stride, stride, stream);
        if (rc) return rc;
        PMT_HIP_CHECK(hipEventRecord(c->computed, s));
        PMT_HIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->computed, 0));
        if (int rc2 = exchange_range(c, local, gathered, per_rank, stride, lo, hi, c->comm_stream)) return rc2;
    }
    PMT_HIP_CHECK(hipEventRecord(c->gathered, c->comm_stream));
    PMT_HIP_CHECK(hipStreamWaitEvent(s, c->gathered, 0));
    return PMT_OK;
}
