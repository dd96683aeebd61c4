// Plan = device buffers + a tape of recorded launches (the native stand-in for the reference's
// LazyExpression/FunctionWrapper evaluation loop, src/lazyexpression.jl:50-61 and
// src/FunctionWrappersQuickFix.jl:108-126, driven by update!(m::Model) src/model.jl:132-143).
//
// The host (Python here, Julia in INTEGRATION.md) analyses the lazy-expression DAG once, sizes every
// `dest` buffer (↔ dest = deepcopy(expr()), src/lazyexpression.jl:202,230,243) and issues the pmt_*_f64
// calls with the plan's recording handle as `stream`; those calls are validated immediately and stored
// as closures.  pmt_plan_update() replays the closures on the plan's HIP stream — no allocation, no
// host-side term bookkeeping (the reference's @allocated == 0 contract) — or, once
// pmt_plan_instantiate_graph() has captured them, launches one hipGraph.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <memory>

#include "common.h"
#include "dma.h"

namespace pmt {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int code, const std::string &msg) { g_last_error = msg; return code; }

}  // namespace pmt

struct pmt_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    bool recording = false;
    std::vector<void *> allocations;
    size_t bytes = 0;
    std::vector<pmt::Launch> tape;    // as recorded, one entry per recorded call
    // what a replay executes: the tape, with every run of consecutive small nodes (small.hip) replaced by one interpreter launch;
    // rebuilt by pmt_plan_end_record / pmt_plan_set_fusion
    std::vector<pmt::Launch> exec;
    std::vector<char> exec_lanes;
    std::vector<int> node_of;         // per tape entry: index into `nodes`, or -1 (an entry only its closure can execute)
    std::vector<pmt::SmallNode> nodes;
    bool fusion = true;
    int fused_groups = 0, fused_nodes = 0, fused_phases = 0, fused_workgroups = 0;
    int *barrier_error = nullptr;            // page-locked word a run on several workgroups stores 1 into when its grid barrier times out
    std::vector<std::pair<void *, size_t>> fused_tables;      // node tables + barrier words of the fused runs: replaced as a whole by build_exec
    hipEvent_t multi_done = nullptr;         // recorded behind this plan's latest launch on several workgroups (device_multi_registry)
    bool multi_pending = false;
    bool single_workgroup_runs = false;      // a graph replays its runs with ONE workgroup (the grid barrier's base is a kernel argument)
    std::vector<char> lanes;          // per tape entry: 0 = the plan's stream, 1 = the side lane, 2 = the FRONT of the side lane, 3 = the front of
                                      // the side lane WITHOUT the fork from the plan's stream (pmt_plan_set_lane)
    char record_lane = 0;
    hipEvent_t lane_fork = nullptr, lane_join = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    char recording_tag = 0;   // &recording_tag is the recording handle
    // staged (overlapped) uploads of host-updated Parameter values: a copy stream of its own and two events
    hipStream_t copy_stream = nullptr;
    // two SLOTS (pmt_plan_stage_slot): the values of update k+1 travel into slot (k+1) % 2 while the commits of update k still read slot k % 2
    hipEvent_t staged[2] = {nullptr, nullptr};     // recorded on the copy stream behind every staged upload of the slot
    hipEvent_t consumed[2] = {nullptr, nullptr};   // recorded on the plan stream behind the slot's commits (its staging buffers may be overwritten after it)
    bool consumed_recorded[2] = {false, false};
    // commits that went to the SIDE stream (pmt_plan_commit_lane): their own event per slot, which the copy stream waits for as well
    hipEvent_t consumed_side[2] = {nullptr, nullptr};
    bool consumed_side_recorded[2] = {false, false};
    bool side_commits = false;         // since the last pmt_plan_staging_consumed
    int commit_lane = 0;
    int slot = 0;
    // pmt_plan_alloc zero-fills on the plan's stream; a staged upload into a fresh buffer must not overtake that fill on the copy stream
    hipEvent_t alloc_done = nullptr;
    bool alloc_pending = false;
    // recorded fetches (pmt_plan_record_fetch): one ordering event per entry
    std::vector<hipEvent_t> fetch_events;
    bool no_graph = false;            // the tape holds an entry whose replay has host-side effects (a host delivery): launches only
};

namespace pmt { int small_plan_occupancy(); }

// A fused run on several workgroups waits at a grid barrier: every workgroup of its grid must be ON the chip.  One such grid is at most
// `multi_workgroup_limit` workgroups (half of what the stream's CUs hold of this kernel); two of them at once — two plans replaying on
// two streams, host threads — could still each hold part of a small partition (CPX: 32 CUs) and wait for the rest until the 2 s bound.
// The library therefore keeps at most ONE multi-workgroup run in flight per device: the plans of a device that own such runs are
// registered here, a launch on several workgroups is made under the device's mutex, and when another registered plan's latest such launch
// has not completed yet (hipEventQuery), THIS launch goes out on one workgroup (same node table, workgroup barriers only: same results).
// Kernels of other processes or libraries are beyond this registry: the clamp, the bounded wait and the error word cover those.
namespace {
struct MultiRegistry { std::mutex mu; std::vector<pmt_plan *> plans; };
MultiRegistry &multi_registry(int device) {
    static std::mutex mu;
    static std::unordered_map<int, std::unique_ptr<MultiRegistry>> regs;
    std::lock_guard<std::mutex> lock(mu);
    auto &r = regs[device];
    if (!r) r.reset(new MultiRegistry);
    return *r;
}
void multi_unregister(pmt_plan *plan) {
    MultiRegistry &r = multi_registry(plan->device);
    std::lock_guard<std::mutex> lock(r.mu);
    r.plans.erase(std::remove(r.plans.begin(), r.plans.end(), plan), r.plans.end());
}
// workgroups of 1024 threads the CUs this stream may use hold at once, halved
int multi_workgroup_limit(pmt_plan *plan) {
    int cus = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, plan->device) == hipSuccess) cus = prop.multiProcessorCount; else (void)hipGetLastError();
    uint32_t mask[16] = {0};
    if (hipExtStreamGetCUMask(plan->stream, 16, mask) == hipSuccess) {
        int bits = 0;
        for (uint32_t m : mask) bits += __builtin_popcount(m);
        if (bits > 0 && (cus == 0 || bits < cus)) cus = bits;
    } else (void)hipGetLastError();
    per_cu = pmt::small_plan_occupancy();
    return std::max(1, cus * std::max(1, per_cu) / 2);
}
// under the registry's mutex: may this plan launch a run on several workgroups now?
bool multi_may_launch(MultiRegistry &r, pmt_plan *plan) {
    for (pmt_plan *other : r.plans) {
        if (other == plan || !other->multi_pending) continue;
        const hipError_t e = hipEventQuery(other->multi_done);
        if (e == hipSuccess) { other->multi_pending = false; continue; }
        (void)hipGetLastError();
        return false;
    }
    return true;
}
int check_barrier_error(pmt_plan *plan) {
    if (plan->barrier_error && __atomic_exchange_n(plan->barrier_error, 0, __ATOMIC_ACQ_REL))
        return pmt::fail(PMT_HIP_ERROR, "small plan: a workgroup waited 2 s at the grid barrier of a fused run (its workgroups were not all running); "
                                        "the outputs of this re-evaluation are invalid");
    return PMT_OK;
}
}  // namespace


namespace pmt {

static std::mutex g_mu;
static std::unordered_map<void *, pmt_plan *> g_recording;   // recording handle -> plan

// the plan whose recording handle `stream` is, or null for an ordinary HIP stream
static pmt_plan *recording_plan(void *stream) {
    if (!stream) return nullptr;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_recording.find(stream);
    return it != g_recording.end() ? it->second : nullptr;
}

// Contract check of pmt_quad_gram_f64 / _csc: the canonical order of the output IS the position order, so `xvar` must be strictly
// increasing (include/parametron_hip.h).  The indices are static, so when the call is RECORDED into a plan they are read back and
// checked once, at record time (setup, not the solve path); an immediate call is checked only in -DPMT_DEBUG_CHECKS builds (it costs a
// stream synchronisation per call).
int check_strictly_increasing(const int64_t *xvar_dev, int64_t n, void *stream) {
    if (n < 2 || !xvar_dev) return PMT_OK;
    pmt_plan *plan = recording_plan(stream);
    hipStream_t s;
    if (plan) {
        s = plan->stream;
    } else {
#ifdef PMT_DEBUG_CHECKS
        s = reinterpret_cast<hipStream_t>(stream);
#else
        return PMT_OK;
#endif
    }
    std::vector<int64_t> h((size_t)n);
    PMT_HIP_CHECK(hipStreamSynchronize(s));                   // the indices may still be on their way (asynchronous upload on this stream)
    PMT_HIP_CHECK(hipMemcpy(h.data(), xvar_dev, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i + 1 < n; ++i)
        if (h[(size_t)i] >= h[(size_t)i + 1])
            return fail(PMT_INVALID_ARGUMENT, "quad_gram: xvar must be strictly increasing (xvar[" + std::to_string(i) + "] = " +
                        std::to_string(h[(size_t)i]) + " >= xvar[" + std::to_string(i + 1) + "] = " + std::to_string(h[(size_t)i + 1]) +
                        "); use the literal objective or canonicalize on the device for other variable orders");
    return PMT_OK;
}

bool is_recording_handle(void *stream) { return recording_plan(stream) != nullptr; }

// a recorded call whose replay does more than enqueue work on the stream (arms signals, submits copy-engine transfers) cannot be captured
void mark_no_graph(void *stream) {
    if (pmt_plan *plan = recording_plan(stream)) plan->no_graph = true;
}

int dispatch(void *stream, Launch launch) {
    if (stream) {
        pmt_plan *plan = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            auto it = g_recording.find(stream);
            if (it != g_recording.end()) plan = it->second;
        }
        if (plan) {
            if (!plan->recording) return fail(PMT_STATE_ERROR, "plan is not recording (call pmt_plan_begin_record first)");
            plan->tape.push_back(std::move(launch));
            plan->lanes.push_back(plan->record_lane);
            plan->node_of.push_back(-1);
            return PMT_OK;
        }
    }
    return launch(reinterpret_cast<hipStream_t>(stream));
}

int dispatch(void *stream, Launch launch, const SmallNode &node) {
    pmt_plan *plan = recording_plan(stream);
    if (plan && plan->recording) {
        plan->nodes.push_back(node);
        const int idx = (int)plan->nodes.size() - 1;
        if (int rc = dispatch(stream, std::move(launch))) { plan->nodes.pop_back(); return rc; }
        plan->node_of.back() = idx;
        if (node.seed_host) plan->no_graph = true;          // a seed read from a host word at every replay cannot be captured
        return PMT_OK;
    }
    return dispatch(stream, std::move(launch));
}

}  // namespace pmt

// ---- per-kernel timing ---------------------------------------------------------------------------------
namespace pmt {

struct ProfRecord { const char *name; hipEvent_t e0, e1; };
// The switch is an atomic (read by every launch of every thread); the filter, the record list and the event pool are only touched under
// g_mu — two plans on two host threads may launch with profiling on (tests/test_gpu_hardening.py).
static std::atomic<bool> g_prof_on{false};
static std::string g_prof_filter;               // non-empty: only kernels whose name contains it are bracketed
static std::vector<ProfRecord> g_prof_records;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {                // caller holds g_mu
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}

ProfScope::ProfScope(const char *name, hipStream_t s) : name_(name), s_(s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return; }
    if (st != hipStreamCaptureStatusNone) return;          // never time inside a graph capture
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_prof_filter.empty() && !strstr(name, g_prof_filter.c_str())) return;
        e0_ = prof_event(); e1_ = prof_event();
    }
    if (e0_ && e1_) (void)hipEventRecord(e0_, s);
}
ProfScope::~ProfScope() {
    if (!e0_ || !e1_) return;
    (void)hipEventRecord(e1_, s_);
    std::lock_guard<std::mutex> lock(g_mu);
    g_prof_records.push_back({name_, e0_, e1_});
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_profile_filter(const char *substring) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_prof_filter = substring ? substring : "";
    return PMT_OK;
}

extern "C" int pmt_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &r : g_prof_records) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
    g_prof_records.clear();
    g_prof_on.store(on != 0);
    return PMT_OK;
}

// Waits for every timed launch, then writes one line per kernel: "<name>\t<count>\t<total_ms>\t<min_ms>\t<max_ms>\n".
// Returns the number of bytes needed (excluding the terminating NUL); truncates to `cap`.
extern "C" int64_t pmt_profile_report(char *buf, size_t cap) {
    std::vector<ProfRecord> recs;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        recs = g_prof_records;
    }
    struct Agg { int64_t n = 0; double tot = 0, mn = 1e300, mx = 0; };
    std::vector<std::pair<std::string, Agg>> aggs;
    for (auto &r : recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        Agg *a = nullptr;
        for (auto &p : aggs) if (p.first == r.name) { a = &p.second; break; }
        if (!a) { aggs.emplace_back(r.name, Agg()); a = &aggs.back().second; }
        a->n++; a->tot += ms; a->mn = std::min(a->mn, (double)ms); a->mx = std::max(a->mx, (double)ms);
    }
    std::string out;
    char line[512];
    for (auto &p : aggs) {
        snprintf(line, sizeof line, "%s\t%lld\t%.6f\t%.6f\t%.6f\n", p.first.c_str(), (long long)p.second.n, p.second.tot, p.second.mn, p.second.mx);
        out += line;
    }
    if (buf && cap) {
        size_t k = std::min(cap - 1, out.size());
        memcpy(buf, out.data(), k);
        buf[k] = 0;
    }
    return (int64_t)out.size();
}

extern "C" const char *pmt_last_error(void) { return g_last_error.c_str(); }
extern "C" int pmt_version(void) { return 100; }
extern "C" int pmt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

namespace pmt {
hipStream_t side_stream_of(hipStream_t s);
void retain_side_stream(hipStream_t s);
void release_side_stream(hipStream_t s);
int fetch_async(hipStream_t s, hipStream_t after, hipEvent_t order_event, void *host_dst, const void *device_src, size_t bytes, FetchState *st, FetchRect r);
int fetch_fence(hipStream_t s);
void replay_begin(hipStream_t s);
int replay_end(hipStream_t s);
int fetch_synchronize(hipStream_t s);
}

extern "C" int pmt_plan_create(int device, void *stream, pmt_plan **out) {
    PMT_REQUIRE(out, PMT_INVALID_ARGUMENT, "plan_create: null out");
    PMT_HIP_CHECK(hipSetDevice(device));
    pmt_plan *p = new pmt_plan();
    p->device = device;
    if (stream) {
        p->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete p; return fail(PMT_HIP_ERROR, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
        p->owns_stream = true;
    }
    {
        std::lock_guard<std::mutex> lock(g_mu);
        g_recording[&p->recording_tag] = p;
    }
    pmt::retain_side_stream(p->stream);
    *out = p;
    return PMT_OK;
}


extern "C" int pmt_plan_destroy(pmt_plan *plan) {
    if (!plan) return PMT_OK;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        g_recording.erase(&plan->recording_tag);
    }
    (void)hipSetDevice(plan->device);
    (void)hipStreamSynchronize(plan->stream);
    // copy-engine transfers of this plan's tape still in flight read its buffers and count on its signals: drained before either goes away,
    // also when other plans keep the stream's auxiliary streams alive
    (void)pmt::fetch_synchronize(plan->stream);
    pmt::release_side_stream(plan->stream);
    if (plan->copy_stream) { (void)hipStreamSynchronize(plan->copy_stream); (void)hipStreamDestroy(plan->copy_stream); }
    for (int i = 0; i < 2; ++i) {
        if (plan->staged[i]) (void)hipEventDestroy(plan->staged[i]);
        if (plan->consumed[i]) (void)hipEventDestroy(plan->consumed[i]);
    }
    if (plan->alloc_done) (void)hipEventDestroy(plan->alloc_done);
    for (hipEvent_t e : plan->fetch_events) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (plan->consumed_side[i]) (void)hipEventDestroy(plan->consumed_side[i]);
    if (plan->lane_fork) (void)hipEventDestroy(plan->lane_fork);
    if (plan->lane_join) (void)hipEventDestroy(plan->lane_join);
    if (plan->graph_exec) (void)hipGraphExecDestroy(plan->graph_exec);
    if (plan->graph) (void)hipGraphDestroy(plan->graph);
    for (void *p : plan->allocations) (void)hipFree(p);
    for (auto &t : plan->fused_tables) (void)hipFree(t.first);
    multi_unregister(plan);
    if (plan->multi_done) (void)hipEventDestroy(plan->multi_done);
    if (plan->barrier_error) (void)hipHostFree(plan->barrier_error);
    if (plan->owns_stream) (void)hipStreamDestroy(plan->stream);
    delete plan;
    return PMT_OK;
}

extern "C" void *pmt_plan_stream(pmt_plan *plan) { return plan ? plan->stream : nullptr; }
extern "C" void *pmt_plan_recording_stream(pmt_plan *plan) { return plan ? &plan->recording_tag : nullptr; }
extern "C" size_t pmt_plan_bytes_allocated(const pmt_plan *plan) { return plan ? plan->bytes : 0; }
extern "C" int64_t pmt_plan_tape_length(const pmt_plan *plan) { return plan ? (int64_t)plan->tape.size() : 0; }

extern "C" int pmt_plan_alloc(pmt_plan *plan, size_t bytes, void **out_device_ptr) {
    PMT_REQUIRE(plan && out_device_ptr, PMT_INVALID_ARGUMENT, "plan_alloc: null argument");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    void *p = nullptr;
    size_t sz = bytes ? bytes : 16;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PMT_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    PMT_HIP_CHECK(hipMemsetAsync(p, 0, sz, plan->stream));
    if (!plan->alloc_done) PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->alloc_done, hipEventDisableTiming));
    PMT_HIP_CHECK(hipEventRecord(plan->alloc_done, plan->stream));
    plan->alloc_pending = true;
    plan->allocations.push_back(p);
    plan->bytes += sz;
    *out_device_ptr = p;
    return PMT_OK;
}

// page-locked host memory for the MOI function buffers: D2H copies into it run at PCIe rate and are truly asynchronous
extern "C" int pmt_host_alloc(size_t bytes, void **out_host_ptr) {
    PMT_REQUIRE(out_host_ptr, PMT_INVALID_ARGUMENT, "host_alloc: null argument");
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PMT_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    memset(p, 0, bytes ? bytes : 16);
    *out_host_ptr = p;
    return PMT_OK;
}

extern "C" int pmt_host_free(void *host_ptr) {
    if (!host_ptr) return PMT_OK;
    PMT_HIP_CHECK(hipHostFree(host_ptr));
    return PMT_OK;
}

extern "C" int pmt_plan_upload(pmt_plan *plan, void *device_dst, const void *host_src, size_t bytes) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_upload: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(device_dst && host_src, PMT_INVALID_ARGUMENT, "plan_upload: null pointer");
    PMT_HIP_CHECK(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, plan->stream));
    return PMT_OK;
}

extern "C" int pmt_plan_zero(pmt_plan *plan, void *device_dst, size_t bytes) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_zero: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(device_dst, PMT_INVALID_ARGUMENT, "plan_zero: null pointer");
    PMT_HIP_CHECK(hipMemsetAsync(device_dst, 0, bytes, plan->stream));
    return PMT_OK;
}

extern "C" int pmt_plan_fetch(pmt_plan *plan, void *host_dst, const void *device_src, size_t bytes) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_fetch: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(host_dst && device_src, PMT_INVALID_ARGUMENT, "plan_fetch: null pointer");
    PMT_HIP_CHECK(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, plan->stream));
    return PMT_OK;
}

// strided (pitched) copies: a column-major matrix whose device copy has a padded leading dimension
extern "C" int pmt_plan_upload_2d(pmt_plan *plan, void *device_dst, size_t dst_pitch, const void *host_src, size_t src_pitch, size_t width_bytes,
                                  size_t height) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_upload_2d: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (width_bytes == 0 || height == 0) return PMT_OK;
    PMT_REQUIRE(device_dst && host_src && dst_pitch >= width_bytes && src_pitch >= width_bytes, PMT_INVALID_ARGUMENT, "plan_upload_2d: bad argument");
    PMT_HIP_CHECK(hipMemcpy2DAsync(device_dst, dst_pitch, host_src, src_pitch, width_bytes, height, hipMemcpyHostToDevice, plan->stream));
    return PMT_OK;
}

extern "C" int pmt_plan_fetch_2d(pmt_plan *plan, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes,
                                 size_t height) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_fetch_2d: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (width_bytes == 0 || height == 0) return PMT_OK;
    PMT_REQUIRE(host_dst && device_src && dst_pitch >= width_bytes && src_pitch >= width_bytes, PMT_INVALID_ARGUMENT, "plan_fetch_2d: bad argument");
    PMT_HIP_CHECK(hipMemcpy2DAsync(host_dst, dst_pitch, device_src, src_pitch, width_bytes, height, hipMemcpyDeviceToHost, plan->stream));
    return PMT_OK;
}

// ---- staged uploads: host-updated Parameters without the serial PCIe copy (src/parameter.jl:88,101-102) ------------
// The reference's `Parameter(model, val=buf)` is a host buffer the user overwrites between solves; update!() reads it when the Parameter
// is evaluated.  Uploading it on the plan's stream puts the PCIe copy serially in front of the kernels (config 2 with host-updated A, b,
// C, d: 151 MB = 2.4 ms before 1.25 ms of kernels).  A STAGED upload goes through a copy stream into a second device buffer instead and
// can therefore run while the previous re-evaluation's kernels are still busy; the next update commits it with a device-to-device copy
// (or the transposition / permutation kernel the Parameter needs anyway) on the plan's stream, ordered by events:
//     copy stream:  [wait consumed(k-1)] H2D values(k) -> staging   [record staged]
//     plan stream:  [wait staged] staging -> parameter buffer [record consumed]   kernels(k) ...
static int ensure_copy_stream(pmt_plan *plan) {
    if (plan->copy_stream) return PMT_OK;
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    PMT_HIP_CHECK(hipStreamCreateWithFlags(&plan->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->staged[i], hipEventDisableTiming));
        PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->consumed[i], hipEventDisableTiming));
        PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->consumed_side[i], hipEventDisableTiming));
    }
    return PMT_OK;
}

static int stage_prologue(pmt_plan *plan) {
    if (int rc = ensure_copy_stream(plan)) return rc;
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    // a staging buffer allocated just now is still being zero-filled on the plan's stream: the copy must come after that
    if (plan->alloc_pending) {
        PMT_HIP_CHECK(hipStreamWaitEvent(plan->copy_stream, plan->alloc_done, 0));
        plan->alloc_pending = false;
    }
    // the slot's staging buffers may still be being read by the commits of the update that used the slot last
    if (plan->consumed_recorded[plan->slot]) PMT_HIP_CHECK(hipStreamWaitEvent(plan->copy_stream, plan->consumed[plan->slot], 0));
    if (plan->consumed_side_recorded[plan->slot]) PMT_HIP_CHECK(hipStreamWaitEvent(plan->copy_stream, plan->consumed_side[plan->slot], 0));
    return PMT_OK;
}

extern "C" int pmt_plan_stage_upload(pmt_plan *plan, void *device_staging, const void *host_src, size_t bytes) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_stage_upload: null plan");
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(device_staging && host_src, PMT_INVALID_ARGUMENT, "plan_stage_upload: null pointer");
    if (int rc = stage_prologue(plan)) return rc;
    PMT_HIP_CHECK(hipMemcpyAsync(device_staging, host_src, bytes, hipMemcpyHostToDevice, plan->copy_stream));
    PMT_HIP_CHECK(hipEventRecord(plan->staged[plan->slot], plan->copy_stream));
    return PMT_OK;
}

extern "C" int pmt_plan_stage_upload_2d(pmt_plan *plan, void *device_staging, size_t dst_pitch, const void *host_src, size_t src_pitch,
                                        size_t width_bytes, size_t height) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_stage_upload_2d: null plan");
    if (width_bytes == 0 || height == 0) return PMT_OK;
    PMT_REQUIRE(device_staging && host_src && dst_pitch >= width_bytes && src_pitch >= width_bytes, PMT_INVALID_ARGUMENT, "plan_stage_upload_2d: bad argument");
    if (int rc = stage_prologue(plan)) return rc;
    PMT_HIP_CHECK(hipMemcpy2DAsync(device_staging, dst_pitch, host_src, src_pitch, width_bytes, height, hipMemcpyHostToDevice, plan->copy_stream));
    PMT_HIP_CHECK(hipEventRecord(plan->staged[plan->slot], plan->copy_stream));
    return PMT_OK;
}

// plan stream: wait for every staged upload issued so far; the caller then enqueues whatever consumes the staging buffers on the plan's
// stream (pmt_plan_commit_staged, or a transposition / permutation kernel) and finishes with pmt_plan_staging_consumed
// the stream the commits of the current commit lane go to: the plan's stream, or (lane 1) the side stream its side-lane entries run on
static hipStream_t commit_stream(pmt_plan *plan) {
    if (plan->commit_lane == 1)
        if (hipStream_t side = pmt::side_stream_of(plan->stream)) return side;
    return plan->stream;
}

extern "C" int pmt_plan_wait_staged(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_wait_staged: null plan");
    if (!plan->copy_stream) return PMT_OK;
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    PMT_HIP_CHECK(hipStreamWaitEvent(commit_stream(plan), plan->staged[plan->slot], 0));
    return PMT_OK;
}

extern "C" int pmt_plan_commit_staged(pmt_plan *plan, void *device_dst, const void *device_staging, size_t bytes) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_commit_staged: null plan");
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(device_dst && device_staging, PMT_INVALID_ARGUMENT, "plan_commit_staged: null pointer");
    if (int rc = pmt_plan_wait_staged(plan)) return rc;
    hipStream_t target = commit_stream(plan);
    PMT_HIP_CHECK(hipMemcpyAsync(device_dst, device_staging, bytes, hipMemcpyDeviceToDevice, target));
    if (target != plan->stream) plan->side_commits = true;
    return PMT_OK;
}

// Lane of the commits that follow (0 = the plan's stream, the default; 1 = its side stream).  A Parameter that only side-lane entries of the
// tape read (pmt_plan_set_lane) can be committed on the side stream: the plan's stream — the contraction of a least-squares objective — then
// does not wait for the upload of, say, the constraint data of this solve.  The caller guarantees that nothing on the plan's stream reads
// the Parameter.
extern "C" int pmt_plan_commit_lane(pmt_plan *plan, int lane) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_commit_lane: null plan");
    PMT_REQUIRE(lane == 0 || lane == 1, PMT_INVALID_ARGUMENT, "plan_commit_lane: lane must be 0 or 1");
    plan->commit_lane = lane;
    return PMT_OK;
}

extern "C" int pmt_plan_staging_consumed(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_staging_consumed: null plan");
    if (int rc = ensure_copy_stream(plan)) return rc;
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    PMT_HIP_CHECK(hipEventRecord(plan->consumed[plan->slot], plan->stream));
    plan->consumed_recorded[plan->slot] = true;
    if (plan->side_commits) {
        if (hipStream_t side = pmt::side_stream_of(plan->stream)) {
            PMT_HIP_CHECK(hipEventRecord(plan->consumed_side[plan->slot], side));
            plan->consumed_side_recorded[plan->slot] = true;
        }
        plan->side_commits = false;
    }
    return PMT_OK;
}

// Two staging slots: stage_upload / wait_staged / commit_staged / staging_consumed act on the CURRENT slot.  A host that alternates the slot
// (and its staging buffers) from one update to the next lets the copy of update k+1 start while the commits of update k are still
// reading their staging buffers; a host that never calls this uses slot 0 throughout (one staging buffer per Parameter).
extern "C" int pmt_plan_stage_slot(pmt_plan *plan, int slot) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_stage_slot: null plan");
    PMT_REQUIRE(slot == 0 || slot == 1, PMT_INVALID_ARGUMENT, "plan_stage_slot: slot must be 0 or 1");
    plan->slot = slot;
    return PMT_OK;
}

// host: block until the staged uploads issued so far have left the HOST buffers (which may then be overwritten)
extern "C" int pmt_plan_staged_synchronize(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_staged_synchronize: null plan");
    if (!plan->copy_stream) return PMT_OK;
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    PMT_HIP_CHECK(hipStreamSynchronize(plan->copy_stream));
    return PMT_OK;
}

extern "C" int pmt_plan_synchronize(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_synchronize: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    PMT_HIP_CHECK(hipStreamSynchronize(plan->stream));
    return check_barrier_error(plan);
}

// For callers that wait for the plan's stream by other means (hipStreamSynchronize, an event, a fetch wait of their own): the plan's
// device-side error state behind such a wait — today the time-out word of a fused run's grid barrier.  PMT_OK, or the error (cleared).
extern "C" int pmt_plan_check(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_check: null plan");
    return check_barrier_error(plan);
}

extern "C" int pmt_plan_begin_record(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_begin_record: null plan");
    PMT_REQUIRE(!plan->recording, PMT_STATE_ERROR, "plan is already recording");
    PMT_REQUIRE(!plan->graph_exec, PMT_STATE_ERROR, "plan graph already instantiated");
    plan->recording = true;
    return PMT_OK;
}

namespace pmt {
size_t small_table_bytes(int count);
void small_table_image(const SmallNode *nodes, int count, void *image);
int small_plan_workgroups(int64_t work);
int launch_small_plan(const void *device_table, int count, const uint64_t *const *seed_words, int ndyn, unsigned long long syncmask,
                      unsigned long long narrowmask, int workgroups, unsigned long long *barrier_word, unsigned long long barrier_base, int *barrier_error,
                      long long barrier_bound, hipStream_t s);
void small_plan_masks(const SmallNode *nodes, int count, unsigned long long *syncmask, unsigned long long *narrowmask);
int small_plan_phases(SmallNode *nodes, int count);
int small_max_nodes();
}

// exec := tape, with every run of >= 2 consecutive small nodes on the plan's own lane replaced by one interpreter launch (small.hip).  The
// node tables live in plan-owned device memory, written here once (setup, not the solve path); a rebuild (pmt_plan_set_fusion, a graph's
// single-workgroup form, a re-record) frees the previous tables first, and the new exec replaces the old one only when it is complete.
static int build_exec_into(pmt_plan *plan, std::vector<pmt::Launch> &exec, std::vector<char> &exec_lanes, std::vector<std::pair<void *, size_t>> &tables);
static int build_exec(pmt_plan *plan) {
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (!plan->fused_tables.empty()) {                 // the launches that read the old tables must be through with them
        PMT_HIP_CHECK(hipStreamSynchronize(plan->stream));
        for (auto &t : plan->fused_tables) { (void)hipFree(t.first); plan->bytes -= t.second; }
        plan->fused_tables.clear();
        plan->exec.clear(); plan->exec_lanes.clear();   // (they point into the freed tables)
    }
    std::vector<pmt::Launch> exec;
    std::vector<char> exec_lanes;
    std::vector<std::pair<void *, size_t>> tables;
    const int rc = build_exec_into(plan, exec, exec_lanes, tables);
    if (rc) {                                          // half built: nothing of it is kept, the plan replays its tape unfused
        for (auto &t : tables) (void)hipFree(t.first);
        plan->exec = plan->tape; plan->exec_lanes = plan->lanes;
        plan->fused_groups = 0; plan->fused_nodes = 0; plan->fused_phases = 0; plan->fused_workgroups = 0;
        return rc;
    }
    plan->exec.swap(exec); plan->exec_lanes.swap(exec_lanes);
    for (auto &t : tables) plan->bytes += t.second;
    plan->fused_tables.swap(tables);
    if (plan->fused_workgroups > 1) {
        MultiRegistry &r = multi_registry(plan->device);
        std::lock_guard<std::mutex> lock(r.mu);
        if (std::find(r.plans.begin(), r.plans.end(), plan) == r.plans.end()) r.plans.push_back(plan);
        if (!plan->multi_done) PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->multi_done, hipEventDisableTiming));
    } else multi_unregister(plan);
    return PMT_OK;
}

static int build_exec_into(pmt_plan *plan, std::vector<pmt::Launch> &out_exec, std::vector<char> &out_lanes, std::vector<std::pair<void *, size_t>> &tables) {
    plan->fused_groups = 0; plan->fused_nodes = 0; plan->fused_phases = 0; plan->fused_workgroups = 0;
    const int wg_limit = multi_workgroup_limit(plan);
    const size_t n = plan->tape.size();
    auto small = [&](size_t i) {
        return plan->fusion && plan->node_of[i] >= 0 && plan->lanes[i] == 0 && plan->nodes[(size_t)plan->node_of[i]].work <= pmt::SMALL_NODE_WORK_MAX;
    };
    for (size_t i = 0; i < n;) {
        size_t j = i;
        int64_t work = 0;
        int ndyn = 0;
        while (j < n && small(j)) {
            const pmt::SmallNode &nd = plan->nodes[(size_t)plan->node_of[j]];
            if (work + nd.work > pmt::SMALL_GROUP_WORK_MAX && j > i) break;
            if (nd.seed_host && ndyn == pmt::SMALL_MAX_DYN) break;
            if ((int)(j - i) == pmt::small_max_nodes()) break;
            work += nd.work;
            ndyn += nd.seed_host ? 1 : 0;
            ++j;
        }
        if (j - i < 2) {
            out_exec.push_back(plan->tape[i]);
            out_lanes.push_back(plan->lanes[i]);
            ++i;
            continue;
        }
        const int count = (int)(j - i);
        std::vector<pmt::SmallNode> group;
        std::vector<const uint64_t *> words;
        for (size_t k = i; k < j; ++k) {
            pmt::SmallNode nd = plan->nodes[(size_t)plan->node_of[k]];
            nd.dyn = -1;
            if (nd.seed_host) { nd.dyn = (int)words.size(); words.push_back(nd.seed_host); }
            group.push_back(nd);
        }
        plan->fused_phases += pmt::small_plan_phases(group.data(), count);
        std::vector<char> image(pmt::small_table_bytes(count));
        pmt::small_table_image(group.data(), count, image.data());
        // the table, and behind it (64-byte aligned) the run's grid-barrier counter + its time-out word, zeroed once
        const size_t table_bytes = (image.size() + 63) / 64 * 64;
        void *table = nullptr;
        hipError_t e = hipMalloc(&table, table_bytes + 64);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(PMT_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        tables.emplace_back(table, table_bytes + 64);
        PMT_HIP_CHECK(hipMemcpy(table, image.data(), image.size(), hipMemcpyHostToDevice));
        PMT_HIP_CHECK(hipMemset(static_cast<char *>(table) + table_bytes, 0, 64));
        unsigned long long syncmask = 0, narrowmask = 0;
        pmt::small_plan_masks(group.data(), count, &syncmask, &narrowmask);
        int64_t group_work = 0;
        for (const pmt::SmallNode &nd : group) group_work += nd.work;
        const int wgs = plan->single_workgroup_runs ? 1 : std::min(wg_limit, pmt::small_plan_workgroups(group_work));
        unsigned long long *bar = reinterpret_cast<unsigned long long *>(static_cast<char *>(table) + table_bytes);
        if (wgs > 1 && !plan->barrier_error) {
            PMT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&plan->barrier_error), 64, hipHostMallocDefault));
            memset(plan->barrier_error, 0, 64);
        }
        int *barrier_error = plan->barrier_error;
        const unsigned long long per_launch = (unsigned long long)__builtin_popcountll(syncmask) * (unsigned long long)wgs;
        // arrivals at the run's counter so far = the next launch's base (a launch on ONE workgroup never touches the counter)
        std::shared_ptr<unsigned long long> arrivals = std::make_shared<unsigned long long>(0);
        plan->fused_workgroups = std::max(plan->fused_workgroups, wgs);
        out_exec.push_back([=](hipStream_t s) {
            if (wgs == 1) return pmt::launch_small_plan(table, count, words.data(), (int)words.size(), syncmask, narrowmask, 1, bar, 0, barrier_error, 0, s);
            MultiRegistry &reg = multi_registry(plan->device);
            std::lock_guard<std::mutex> lock(reg.mu);                 // one multi-workgroup run in flight per device (see multi_registry)
            const bool multi = multi_may_launch(reg, plan) && !(pmt::dma::fault_injection() & 4);
            unsigned long long base = *arrivals;
            long long bound = 200000000LL;                            // 2 s of 100 MHz ticks
            // test hook (pmt_set_fault_injection(2)): every barrier of this launch waits for one arrival more than there will be, for 20 ms
            if (multi && (pmt::dma::fault_injection() & 2)) { base += 1; bound = 2000000LL; }
            const int rc = pmt::launch_small_plan(table, count, words.data(), (int)words.size(), syncmask, narrowmask, multi ? wgs : 1, bar, base, barrier_error, bound, s);
            if (!rc && multi) {
                *arrivals += per_launch;                              // (a launch that was not enqueued does not arrive at the counter)
                if (reg.plans.size() > 1 && plan->multi_done && hipEventRecord(plan->multi_done, s) == hipSuccess) plan->multi_pending = true;
            }
            return rc;
        });
        out_lanes.push_back(0);
        plan->fused_groups += 1;
        plan->fused_nodes += count;
        i = j;
    }
    return PMT_OK;
}

extern "C" int pmt_plan_end_record(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_end_record: null plan");
    PMT_REQUIRE(plan->recording, PMT_STATE_ERROR, "plan is not recording");
    plan->recording = false;
    plan->record_lane = 0;
    return build_exec(plan);
}

extern "C" int pmt_plan_set_fusion(pmt_plan *plan, int on) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_set_fusion: null plan");
    PMT_REQUIRE(!plan->recording, PMT_STATE_ERROR, "plan_set_fusion: the plan is recording");
    PMT_REQUIRE(!plan->graph_exec, PMT_STATE_ERROR, "plan_set_fusion: the plan's graph is already instantiated");
    plan->fusion = on != 0;
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    PMT_HIP_CHECK(hipStreamSynchronize(plan->stream));
    return build_exec(plan);
}

extern "C" int pmt_plan_fused_phases(const pmt_plan *plan) { return plan ? plan->fused_phases : 0; }
extern "C" int pmt_plan_fused_workgroups(const pmt_plan *plan) { return plan ? plan->fused_workgroups : 0; }

extern "C" int pmt_plan_fused(const pmt_plan *plan, int *groups, int *nodes, int64_t *exec_length) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_fused: null plan");
    if (groups) *groups = plan->fused_groups;
    if (nodes) *nodes = plan->fused_nodes;
    if (exec_length) *exec_length = (int64_t)plan->exec.size();
    return PMT_OK;
}


// Side-lane entries (pmt_plan_set_lane) only read buffers that were complete BEFORE the replay started (Parameter values) — or the AFFINE
// part of a Gram node recorded before them (the hand-off's q gather): they are queued on the calling stream's side stream, BEHIND that
// node's affine reduction — and write outputs nothing else in the tape reads, so they fork at the top of the replay and join at its end.
// (One side stream, not two: a second low-priority stream per plan cost config 3 its overlapped uploads, 1.27 -> 1.76 ms per step —
// the runtime maps streams onto a handful of hardware queues, and the plan's copy stream ended up sharing one.  profiles/r03_host_delivery.txt)
static int replay(pmt_plan *plan, hipStream_t s) {
    // copies of the previous re-evaluation that are still on the fetch stream read buffers this one is about to overwrite
    if (!plan->fetch_events.empty())
        if (int rc = pmt::fetch_fence(s)) return rc;
    pmt::replay_begin(s);
    struct End { hipStream_t s; int rc = PMT_OK; bool done = false; int finish() { if (!done) { done = true; rc = pmt::replay_end(s); } return rc; } ~End() { finish(); } } end{s};
    hipStream_t side = nullptr;
    bool any = false;
    for (char l : plan->exec_lanes) any |= (l != 0);
    if (any && (side = pmt::side_stream_of(s))) {
        if (!plan->lane_fork) {
            PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->lane_fork, hipEventDisableTiming));
            PMT_HIP_CHECK(hipEventCreateWithFlags(&plan->lane_join, hipEventDisableTiming));
        }
        PMT_HIP_CHECK(hipEventRecord(plan->lane_fork, s));
    }
    bool forked = false;
    // lane 2 first: entries that depend on nothing of this replay (a transfer of Parameter values that are already in place) are queued on
    // the side stream in front of everything the tape puts there — e.g. in front of the Gram node's affine reduction, behind which a
    // lane-1 entry would start ~0.1 ms into the re-evaluation
    // lane 3 before that, without the fork: entries whose inputs are produced ON the side stream (a Parameter committed / regenerated there,
    // pmt_plan_commit_lane / pmt_plan_lane_stream) do not wait for what the plan's stream still has to do before the tape — e.g. the
    // device-side callback of the objective's 134 MB matrix
    bool side_used = false;
    for (size_t i = 0; side && i < plan->exec.size(); ++i)
        if (plan->exec_lanes[i] == 3) {
            side_used = true;
            if (int rc = plan->exec[i](side)) return rc;
        }
    for (size_t i = 0; side && i < plan->exec.size(); ++i) {
        if (plan->exec_lanes[i] != 2) continue;
        if (!forked) { PMT_HIP_CHECK(hipStreamWaitEvent(side, plan->lane_fork, 0)); forked = true; }
        if (int rc = plan->exec[i](side)) return rc;
    }
    for (size_t i = 0; i < plan->exec.size(); ++i) {
        hipStream_t target = s;
        if (side && plan->exec_lanes[i] >= 2) continue;
        if (side && plan->exec_lanes[i]) {
            if (!forked) { PMT_HIP_CHECK(hipStreamWaitEvent(side, plan->lane_fork, 0)); forked = true; }
            target = side;
        }
        int rc = plan->exec[i](target);
        if (rc) return rc;
    }
    if (forked || side_used) {
        PMT_HIP_CHECK(hipEventRecord(plan->lane_join, side));
        PMT_HIP_CHECK(hipStreamWaitEvent(s, plan->lane_join, 0));
    }
    return end.finish();
}

// ---- recorded fetches: results leave for the host while the tape is still running -----------------------------------------------------
// The reference hands its MOI functions to a HOST solver (MOI.set, src/moi_interop.jl:134,171).  pmt_plan_fetch puts a D2H copy on the
// plan's stream, i.e. behind the whole re-evaluation.  A RECORDED fetch is a tape entry: at replay an event is recorded on the stream the
// entry's lane runs on (everything recorded before it on that lane has been enqueued), and the copy goes to the plan's FETCH stream behind
// that event — the constraint block of config 2 (lane 1, finished early) crosses PCIe while the contraction is still busy.
extern "C" int pmt_plan_record_fetch(pmt_plan *plan, void *host_dst, const void *device_src, size_t bytes) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_record_fetch: null plan");
    PMT_REQUIRE(plan->recording, PMT_STATE_ERROR, "plan_record_fetch: the plan is not recording");
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(host_dst && device_src, PMT_INVALID_ARGUMENT, "plan_record_fetch: null pointer");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    hipEvent_t ev = nullptr;
    PMT_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    plan->fetch_events.push_back(ev);
    hipStream_t main = plan->stream;
    std::shared_ptr<pmt::FetchState> st = std::make_shared<pmt::FetchState>();     // the transfer's signals; goes with the tape entry
    plan->tape.push_back([=](hipStream_t target) { return pmt::fetch_async(main, target, ev, host_dst, device_src, bytes, st.get(), pmt::FetchRect{}); });
    plan->lanes.push_back(plan->record_lane);
    plan->node_of.push_back(-1);
    return PMT_OK;
}

// The pitched form: `height` rows of `width_bytes` (a dense matrix block whose device copy is padded, leaving for a column range of the
// solver's stacked constraint matrix — the CSC values of a dense block ARE the Parameter matrix column by column, so they leave straight
// out of the Parameter's buffer, with no kernel in between).
extern "C" int pmt_plan_record_fetch_2d(pmt_plan *plan, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes,
                                        size_t height) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_record_fetch_2d: null plan");
    PMT_REQUIRE(plan->recording, PMT_STATE_ERROR, "plan_record_fetch_2d: the plan is not recording");
    if (width_bytes == 0 || height == 0) return PMT_OK;
    PMT_REQUIRE(host_dst && device_src && dst_pitch >= width_bytes && src_pitch >= width_bytes, PMT_INVALID_ARGUMENT, "plan_record_fetch_2d: bad argument");
    if (dst_pitch == width_bytes && src_pitch == width_bytes) return pmt_plan_record_fetch(plan, host_dst, device_src, width_bytes * height);
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    hipEvent_t ev = nullptr;
    PMT_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    plan->fetch_events.push_back(ev);
    hipStream_t main = plan->stream;
    std::shared_ptr<pmt::FetchState> st = std::make_shared<pmt::FetchState>();
    pmt::FetchRect r;
    r.dst_pitch = dst_pitch; r.src_pitch = src_pitch; r.height = height;
    plan->tape.push_back([=](hipStream_t target) { return pmt::fetch_async(main, target, ev, host_dst, device_src, width_bytes, st.get(), r); });
    plan->lanes.push_back(plan->record_lane);
    plan->node_of.push_back(-1);
    return PMT_OK;
}

// host: block until every copy the plan's fetch stream has been given (recorded fetches, delivered CSC values) has landed
extern "C" int pmt_plan_fetch_synchronize(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_fetch_synchronize: null plan");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (int rc = pmt::fetch_synchronize(plan->stream)) return rc;
    return check_barrier_error(plan);          // (the fetched buffers were written by the replay whose fused runs report here)
}

extern "C" int pmt_plan_set_lane(pmt_plan *plan, int lane) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_set_lane: null plan");
    PMT_REQUIRE(plan->recording, PMT_STATE_ERROR, "plan_set_lane: the plan is not recording");
    PMT_REQUIRE(lane >= 0 && lane <= 3, PMT_INVALID_ARGUMENT, "plan_set_lane: lane must be 0, 1, 2 or 3");
    plan->record_lane = (char)lane;
    return PMT_OK;
}

// the HIP stream a lane's entries are replayed on: 0 the plan's stream, 1 / 2 its side stream (created on first use)
extern "C" int pmt_plan_lane_stream(pmt_plan *plan, int lane, void **out_stream) {
    PMT_REQUIRE(plan && out_stream, PMT_INVALID_ARGUMENT, "plan_lane_stream: null pointer");
    PMT_REQUIRE(lane >= 0 && lane <= 2, PMT_INVALID_ARGUMENT, "plan_lane_stream: lane must be 0, 1 or 2");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    hipStream_t s = lane == 0 ? plan->stream : pmt::side_stream_of(plan->stream);
    PMT_REQUIRE(s || lane == 0, PMT_STATE_ERROR, "plan_lane_stream: the plan's stream has no side stream");
    *out_stream = s;
    return PMT_OK;
}

extern "C" int pmt_plan_update(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_update: null plan");
    PMT_REQUIRE(!plan->recording, PMT_STATE_ERROR, "plan is still recording");
    PMT_HIP_CHECK(hipSetDevice(plan->device));      // the tape's launches go to this plan's device whatever the caller's current device is
    if (plan->graph_exec) {
        PMT_HIP_CHECK(hipGraphLaunch(plan->graph_exec, plan->stream));
        return PMT_OK;
    }
    return replay(plan, plan->stream);
}

extern "C" int pmt_plan_instantiate_graph(pmt_plan *plan) {
    PMT_REQUIRE(plan, PMT_INVALID_ARGUMENT, "plan_instantiate_graph: null plan");
    PMT_REQUIRE(!plan->recording, PMT_STATE_ERROR, "plan is still recording");
    if (plan->graph_exec) return PMT_OK;
    PMT_REQUIRE(plan->fetch_events.empty() && !plan->no_graph, PMT_STATE_ERROR,
                "plan_instantiate_graph: a tape with recorded fetches or a host delivery is replayed as launches (its copies leave the capture)");
    PMT_HIP_CHECK(hipSetDevice(plan->device));
    if (plan->fused_workgroups > 1) {
        // a run on several workgroups counts its grid barriers from a base that is a kernel argument of each launch: a captured launch
        // would replay a stale one — inside a graph the runs are single-workgroup launches (as they all were before round 5)
        PMT_HIP_CHECK(hipStreamSynchronize(plan->stream));
        plan->single_workgroup_runs = true;
        if (int rc = build_exec(plan)) return rc;
    }
    PMT_HIP_CHECK(hipStreamBeginCapture(plan->stream, hipStreamCaptureModeThreadLocal));
    int rc = replay(plan, plan->stream);
    hipError_t e = hipStreamEndCapture(plan->stream, &plan->graph);
    if (rc) { if (plan->graph) { (void)hipGraphDestroy(plan->graph); plan->graph = nullptr; } return rc; }
    if (e != hipSuccess) return fail(PMT_HIP_ERROR, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    PMT_HIP_CHECK(hipGraphInstantiate(&plan->graph_exec, plan->graph, nullptr, nullptr, 0));
    return PMT_OK;
}
