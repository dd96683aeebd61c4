// Copy-engine (SDMA) delivery of results to the host, gated by signals the KERNELS set.
//
// Why not hipMemcpyAsync: measured on this stack (profiles/r03_host_delivery.txt) the HIP runtime bundled with PyTorch-ROCm 7.0 performs a
// device -> page-locked-host hipMemcpyAsync with a blit KERNEL (__amd_rocclr_copyBuffer), /opt/rocm 7.2's with the copy engine; and a
// stream-level "wait until this memory word changes" (hipStreamWaitValue64) is a spinning kernel in both.  Kernels that copy or spin sit on
// the CUs of the persistent contraction and slowed it by 40-60 %.  The HSA runtime underneath both offers what the hardware has:
// hsa_amd_memory_async_copy runs on an SDMA engine and starts when its DEPENDENCY SIGNALS read 0 — and a signal's value is an ordinary
// 64-bit word in memory that a kernel can write.  So a one-thread kernel behind the stage of the contraction that completes a band group
// stores 0 into the group's signal, and the engine ships that group while the matrix cores go on with the next stage; no CU is involved in
// waiting or copying.
//
// The HSA runtime is the one the HIP runtime of this process has already loaded (found with dl_iterate_phdr, opened RTLD_NOLOAD): no second
// runtime, no new dependency at link time.  If it cannot be found, or the device cannot be matched to an HSA agent, dma::get() returns null
// and the callers fall back to kernel copies (deliver.hip).
#include <dlfcn.h>
#include <link.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <hsa/amd_hsa_signal.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include "common.h"
#include "dma.h"

namespace pmt {
namespace dma {

struct Api {
    decltype(&hsa_init) init = nullptr;
    decltype(&hsa_iterate_agents) iterate_agents = nullptr;
    decltype(&hsa_agent_get_info) agent_get_info = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store_relaxed = nullptr;
    decltype(&hsa_signal_load_relaxed) signal_load_relaxed = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait_scacquire = nullptr;
    decltype(&hsa_amd_memory_async_copy) memory_async_copy = nullptr;
    // optional (HSA 1.2+): a chosen SDMA engine per copy, so that two FIFO queues exist instead of one
    decltype(&hsa_amd_memory_async_copy_on_engine) memory_async_copy_on_engine = nullptr;
    decltype(&hsa_amd_memory_copy_engine_status) memory_copy_engine_status = nullptr;
    decltype(&hsa_amd_memory_async_copy_rect) memory_async_copy_rect = nullptr;      // optional: pitched transfers on the engine
};

struct Engine {
    Api api;
    hsa_agent_t gpu{}, cpu{};
    uint32_t queue_engine[2] = {0, 0};      // SDMA engine of queue 0 (recorded fetches) / queue 1 (P's stages); 0: the runtime chooses
};

static std::mutex g_mu;
static std::atomic<int> g_mode{0}, g_fault{0};
int delivery_mode() { return g_mode.load(std::memory_order_relaxed); }
int fault_injection() { return g_fault.load(std::memory_order_relaxed); }
static bool g_tried = false;
static void *g_lib = nullptr;
static Api g_api;
static std::vector<Engine *> g_engines;        // per HIP device (null: no match)
static std::vector<int> g_failed;              // devices whose engine did not pass its self-test
__global__ void signal_store_kernel(int64_t *value);
static bool self_test(Engine *e, int device);

static int find_hsa(struct dl_phdr_info *info, size_t, void *data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "libhsa-runtime64")) {
        *static_cast<std::string *>(data) = info->dlpi_name;
        return 1;
    }
    return 0;
}

static bool load_api() {
    if (g_tried) return g_lib != nullptr;
    g_tried = true;
#ifdef PMT_TUNING
    if (const char *e = getenv("PMT_DMA")) if (e[0] == '0') return false;
#endif
    std::string path;
    dl_iterate_phdr(find_hsa, &path);
    if (path.empty()) return false;
    void *lib = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (!lib) return false;
#define PMT_HSA_SYM(field, name)                                              \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(lib, name)); \
    if (!g_api.field) { dlclose(lib); return false; }
    PMT_HSA_SYM(init, "hsa_init")
    PMT_HSA_SYM(iterate_agents, "hsa_iterate_agents")
    PMT_HSA_SYM(agent_get_info, "hsa_agent_get_info")
    PMT_HSA_SYM(signal_create, "hsa_signal_create")
    PMT_HSA_SYM(signal_destroy, "hsa_signal_destroy")
    PMT_HSA_SYM(signal_store_relaxed, "hsa_signal_store_relaxed")
    PMT_HSA_SYM(signal_load_relaxed, "hsa_signal_load_relaxed")
    PMT_HSA_SYM(signal_wait_scacquire, "hsa_signal_wait_scacquire")
    PMT_HSA_SYM(memory_async_copy, "hsa_amd_memory_async_copy")
#undef PMT_HSA_SYM
    g_api.memory_async_copy_on_engine = reinterpret_cast<decltype(g_api.memory_async_copy_on_engine)>(dlsym(lib, "hsa_amd_memory_async_copy_on_engine"));
    g_api.memory_copy_engine_status = reinterpret_cast<decltype(g_api.memory_copy_engine_status)>(dlsym(lib, "hsa_amd_memory_copy_engine_status"));
    g_api.memory_async_copy_rect = reinterpret_cast<decltype(g_api.memory_async_copy_rect)>(dlsym(lib, "hsa_amd_memory_async_copy_rect"));
    if (g_api.init() != HSA_STATUS_SUCCESS) { dlclose(lib); return false; }     // reference-counted: HIP initialised it long ago
    g_lib = lib;
    return true;
}

struct AgentList { std::vector<hsa_agent_t> gpus, cpus; const Api *api; };
static hsa_status_t collect_agent(hsa_agent_t a, void *data) {
    AgentList *l = static_cast<AgentList *>(data);
    hsa_device_type_t t;
    if (l->api->agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_GPU) l->gpus.push_back(a);
    else if (t == HSA_DEVICE_TYPE_CPU) l->cpus.push_back(a);
    return HSA_STATUS_SUCCESS;
}

static Engine *engine_of(int device);
Engine *get(int device) { return delivery_mode() == 2 ? nullptr : engine_of(device); }
static Engine *engine_of(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!load_api()) return nullptr;
    if (device < 0) return nullptr;
    for (int d : g_failed) if (d == device) return nullptr;          // matched before and failed its self-test
    if ((size_t)device < g_engines.size() && g_engines[(size_t)device]) return g_engines[(size_t)device];
    AgentList l;
    l.api = &g_api;
    if (g_api.iterate_agents(collect_agent, &l) != HSA_STATUS_SUCCESS || l.gpus.empty() || l.cpus.empty()) return nullptr;
    // the HSA agent of HIP device `device`: same PCI domain / bus / device (HIP_VISIBLE_DEVICES renumbers HIP devices only).  On a multi-socket
    // node two GPUs may share bus:device in different domains, so the domain is part of the key; an ambiguous match means no engine (the
    // kernel copies are used) rather than another GPU's.
    int bus = -1, dev = -1, domain = -1;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess ||
        hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipDeviceGetAttribute(&domain, hipDeviceAttributePciDomainID, device) != hipSuccess) { (void)hipGetLastError(); domain = -1; }
    hsa_agent_t gpu{};
    int matches = 0;
    for (hsa_agent_t a : l.gpus) {
        uint32_t bdf = 0, dom = 0;
        if (g_api.agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) continue;
        if ((int)((bdf >> 8) & 0xff) != bus || (int)((bdf >> 3) & 0x1f) != dev) continue;
        const bool have_dom = g_api.agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) == HSA_STATUS_SUCCESS;
        if (have_dom && domain >= 0 && (int)dom != domain) continue;
        gpu = a;
        ++matches;
    }
    const bool found = matches == 1;
    if (!found) return nullptr;
    Engine *e = new Engine();
    e->api = g_api; e->gpu = gpu; e->cpu = l.cpus[0];
    // Two device -> host queues when the runtime lets us pick engines: a transfer whose data appears late (A's values behind this solve's
    // Parameter upload, config 3) then does not hold back P's stages queued behind it, and vice versa.  PCIe is shared either way.
#ifdef PMT_TUNING
    const bool two_queues = !(getenv("PMT_DMA_QUEUES") && getenv("PMT_DMA_QUEUES")[0] == '1');
#else
    const bool two_queues = true;
#endif
    if (two_queues && g_api.memory_async_copy_on_engine && g_api.memory_copy_engine_status) {
        uint32_t mask = 0;
        if (g_api.memory_copy_engine_status(e->cpu, e->gpu, &mask) == HSA_STATUS_SUCCESS) {
            int n = 0;
            for (uint32_t bit = 1; bit && n < 2; bit <<= 1)
                if (mask & bit) e->queue_engine[n++] = bit;
            if (n < 2) e->queue_engine[0] = e->queue_engine[1] = 0;
        }
    }
    // The delivery rests on two things the HSA API does not promise: a signal's value word may be written by a KERNEL, and the copy engine
    // notices that write.  Checked here, once per device, with a real transfer: if it does not behave, there is no engine (kernel copies).
    if (!self_test(e, device)) { delete e; e = nullptr; }
    if (g_engines.size() <= (size_t)device) g_engines.resize((size_t)device + 1, nullptr);
    g_engines[(size_t)device] = e;
    if (!e) g_failed.push_back(device);
    return e;
}

int signal_create(Engine *e, int64_t initial, Signal *out) {
    hsa_signal_t s{};
    if (e->api.signal_create(initial, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return fail(PMT_HIP_ERROR, "hsa_signal_create failed");
    out->handle = s.handle;
    // a signal is an amd_signal_t in system memory every agent can reach; its value is the word the copy engine polls and a kernel may write
    out->value = const_cast<int64_t *>(&reinterpret_cast<amd_signal_t *>(s.handle)->value);
    return PMT_OK;
}

void signal_destroy(Engine *e, Signal s) {
    if (s.handle) (void)e->api.signal_destroy(hsa_signal_t{s.handle});
}

void signal_set(Engine *e, Signal s, int64_t v) { e->api.signal_store_relaxed(hsa_signal_t{s.handle}, v); }

#ifdef PMT_TUNING
static bool dbg() { static const bool on = getenv("PMT_DMA_DEBUG") != nullptr; return on; }
static double dnow() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
#endif

int copy_to_host(Engine *e, void *host_dst, const void *device_src, size_t bytes, const Signal *dep, Signal completion, int queue) {
    hsa_signal_t d{};
    if (dep) d.handle = dep->handle;
#ifdef PMT_TUNING
    if (dbg()) fprintf(stderr, "[dma %.0f] submit %zu bytes dep=%ld done=%ld (done handle %lx)\n", dnow(), bytes, dep ? (long)*dep->value : -99L, (long)*completion.value, (unsigned long)completion.handle);
#endif
    hsa_status_t st;
    const uint32_t eng = e->queue_engine[queue & 1];
    if (eng) {
        st = e->api.memory_async_copy_on_engine(host_dst, e->cpu, device_src, e->gpu, bytes, dep ? 1 : 0, dep ? &d : nullptr, hsa_signal_t{completion.handle},
                                                (hsa_amd_sdma_engine_id_t)eng, false);
        if (st != HSA_STATUS_SUCCESS) {                    // the engine went away (busy, reset): one queue from now on
            e->queue_engine[0] = e->queue_engine[1] = 0;
            st = e->api.memory_async_copy(host_dst, e->cpu, device_src, e->gpu, bytes, dep ? 1 : 0, dep ? &d : nullptr, hsa_signal_t{completion.handle});
        }
    } else {
        st = e->api.memory_async_copy(host_dst, e->cpu, device_src, e->gpu, bytes, dep ? 1 : 0, dep ? &d : nullptr, hsa_signal_t{completion.handle});
    }
    if (st != HSA_STATUS_SUCCESS) return fail(PMT_HIP_ERROR, "hsa_amd_memory_async_copy failed (status " + std::to_string((int)st) + ")");
    return PMT_OK;
}

// A pitched block.  The engine's rectangle copy where the runtime has it; otherwise (or when it refuses the shape) one linear copy per
// row, all counted down on the same completion signal — the caller has set it to 1, so it is raised to `height` first.
int copy_rect_to_host(Engine *e, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes, size_t height,
                      const Signal *dep, Signal completion) {
    hsa_signal_t d{};
    if (dep) d.handle = dep->handle;
    if (e->api.memory_async_copy_rect && width_bytes < ((size_t)1 << 32) && height < ((size_t)1 << 32) && (width_bytes & 3) == 0 &&
        (dst_pitch & 3) == 0 && (src_pitch & 3) == 0) {
        hsa_pitched_ptr_t dp{host_dst, dst_pitch, dst_pitch * height}, sp{const_cast<void *>(device_src), src_pitch, src_pitch * height};
        hsa_dim3_t zero{0, 0, 0}, range{(uint32_t)width_bytes, (uint32_t)height, 1};
        const hsa_status_t st = e->api.memory_async_copy_rect(&dp, &zero, &sp, &zero, &range, e->gpu, hsaDeviceToHost, dep ? 1 : 0, dep ? &d : nullptr,
                                                              hsa_signal_t{completion.handle});
        if (st == HSA_STATUS_SUCCESS) return PMT_OK;
    }
    signal_set(e, completion, (int64_t)height);
    for (size_t r = 0; r < height; ++r)
        if (int rc = copy_to_host(e, static_cast<char *>(host_dst) + r * dst_pitch, static_cast<const char *>(device_src) + r * src_pitch, width_bytes, dep, completion))
            return rc;
    return PMT_OK;
}

// host: until the completion signal has counted down to 0 (every copy that decrements it is done); a negative value is the runtime's error
// report; `timeout_s` bounds the wait (a dependency that is never signalled must not hang the host for good)
int wait(Engine *e, Signal completion, double timeout_s) {
    const uint64_t slice = 2000000000ull;                                       // the timeout argument is in HSA system-clock ticks: wait in slices
    const double t0 = (double)clock() / CLOCKS_PER_SEC;
    timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
#ifdef PMT_TUNING
    if (dbg()) fprintf(stderr, "[dma %.0f] wait on %lx: value %ld\n", dnow(), (unsigned long)completion.handle, (long)*completion.value);
#endif
    // The transfers of a re-evaluation are a couple of milliseconds away at most: poll the signal's word first, as the HIP runtime's own
    // stream synchronisation does by default (a blocked wait wakes up 10-40 us after the engine's interrupt, more on a busy host); block
    // only when the copies are not about to finish
    for (;;) {
        const int64_t v = __atomic_load_n(completion.value, __ATOMIC_ACQUIRE);
        if (v < 0) return fail(PMT_HIP_ERROR, "host delivery: the copy engine reported an error");
        if (v == 0) return PMT_OK;
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        if ((double)(ts.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts.tv_nsec - ts0.tv_nsec) > 6e-3) break;
        __builtin_ia32_pause();
    }
    for (;;) {
        const hsa_signal_value_t v = e->api.signal_wait_scacquire(hsa_signal_t{completion.handle}, HSA_SIGNAL_CONDITION_LT, 1, slice, HSA_WAIT_STATE_BLOCKED);
        if (v < 0) return fail(PMT_HIP_ERROR, "host delivery: the copy engine reported an error");
#ifdef PMT_TUNING
        if (dbg()) fprintf(stderr, "[dma %.0f] wait on %lx returned %ld\n", dnow(), (unsigned long)completion.handle, (long)v);
#endif
        if (v == 0) return PMT_OK;
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        if ((double)(ts.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts.tv_nsec - ts0.tv_nsec) > timeout_s)
            return fail(PMT_HIP_ERROR, "host delivery: timed out waiting for the copy engine (a producer never signalled its data)");
    }
    (void)t0;
}

// one thread: the producers enqueued before this launch on `s` are done -> the copy that depends on `value` may start
__global__ void signal_store_kernel(int64_t *value) {
    __hip_atomic_store(value, (int64_t)0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One 64-byte transfer gated the way every delivery is: submitted with a dependency signal at 1, it must NOT have run after a millisecond; a
// one-thread kernel then stores 0 into the signal's value word and the data must arrive.  false: the engine is not used on this device.
static bool self_test(Engine *e, int device) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    unsigned long long *dsrc = nullptr, *hdst = nullptr;
    hipStream_t st = nullptr;
    Signal dep, done;
    bool ok = hipMalloc(reinterpret_cast<void **>(&dsrc), 64) == hipSuccess && hipHostMalloc(reinterpret_cast<void **>(&hdst), 64, hipHostMallocDefault) == hipSuccess &&
              hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    bool sig = false;
    if (ok) {
        const unsigned long long pattern[8] = {0x70617261ull, 0x6d657472ull, 0x6f6eull, 4, 5, 6, 7, 0x5e1f7e57ull};
        for (int i = 0; i < 8; ++i) hdst[i] = 0;
        ok = hipMemcpy(dsrc, pattern, 64, hipMemcpyHostToDevice) == hipSuccess;
        sig = ok && signal_create(e, 1, &dep) == PMT_OK && signal_create(e, 1, &done) == PMT_OK;
        ok = sig && copy_to_host(e, hdst, dsrc, 64, &dep, done) == PMT_OK;
        if (ok) {
            timespec nap{0, 1000000};
            nanosleep(&nap, nullptr);
            ok = __atomic_load_n(done.value, __ATOMIC_ACQUIRE) == 1 && hdst[7] == 0;      // held back by its dependency
            hipLaunchKernelGGL(signal_store_kernel, dim3(1), dim3(1), 0, st, dep.value);
            const bool launched = hipGetLastError() == hipSuccess;
            // (wait for the transfer in any case: the signals and buffers must not go away under it)
            const bool landed = launched && wait(e, done, 0.5) == PMT_OK;
            if (!launched) signal_set(e, dep, 0), (void)wait(e, done, 0.5);
            ok = ok && landed;
            for (int i = 0; ok && i < 8; ++i) ok = hdst[i] == pattern[i];
        }
    }
    if (sig) { signal_destroy(e, dep); signal_destroy(e, done); }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (dsrc) (void)hipFree(dsrc);
    if (hdst) (void)hipHostFree(hdst);
    (void)hipGetLastError();
    (void)hipSetDevice(prev);
    return ok;
}

int launch_signal_store(Signal s, hipStream_t stream) {
    PMT_LAUNCH(signal_store_kernel, dim3(1), dim3(1), 0, stream, s.value);
    return check_launch("signal_store_kernel");
}

}  // namespace dma
}  // namespace pmt

// Which way results leave for the host (recorded fetches, band-wise deliveries): takes effect from the next replay / call on.
extern "C" int pmt_set_host_delivery(int mode) {
    PMT_REQUIRE(mode >= 0 && mode <= 2, PMT_INVALID_ARGUMENT, "set_host_delivery: mode must be 0 (auto), 1 (copy engine) or 2 (kernel copies)");
    pmt::dma::g_mode.store(mode);
    return PMT_OK;
}

extern "C" int pmt_get_host_delivery(int device, int *out_mode, int *out_copy_engine) {
    if (out_mode) *out_mode = pmt::dma::delivery_mode();
    if (out_copy_engine) {
        *out_copy_engine = pmt::dma::engine_of(device) ? 1 : 0;      // (whatever the mode says)
    }
    return PMT_OK;
}

extern "C" int pmt_set_fault_injection(int what) {
    PMT_REQUIRE(what >= 0 && what <= 7, PMT_INVALID_ARGUMENT, "set_fault_injection: unknown fault");
    pmt::dma::g_fault.store(what);
    return PMT_OK;
}
