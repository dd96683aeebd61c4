// Does hsa_amd_memory_async_copy honour a dependency signal whose value a KERNEL writes, and when does the copy start?
// build: hipcc --offload-arch=gfx950 -O2 tools/hsa_dep_probe.hip -o tools/hsa_dep_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/amd_hsa_signal.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define HK(x) do { hsa_status_t e_ = (x); if (e_ != HSA_STATUS_SUCCESS) { printf("%s -> %d\n", #x, (int)e_); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::vector<hsa_agent_t> gpus, cpus;
static hsa_status_t cb(hsa_agent_t a, void *) { hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t); (t == HSA_DEVICE_TYPE_GPU ? gpus : cpus).push_back(a); return HSA_STATUS_SUCCESS; }

// fills data with `tag`, waits `delay` ticks (100 MHz), then stores 0 to *sig (plain system-scope store), keeps running `tail` ticks
__global__ void producer(double *data, int n, double tag, long long delay, long long tail, long long *sig, int mode) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(data + i, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) {
        if (mode == 0) __hip_atomic_store(sig, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(sig, 0ll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    t0 = wall_clock64();
    while (wall_clock64() - t0 < tail) __builtin_amdgcn_s_sleep(10);
}

// fires sig[order[i]] at (i + 1) * step ticks
__global__ void multi_producer(long long **sig, const int *order, int nsig, long long step) {
    long long t0 = wall_clock64();
    for (int i = 0; i < nsig; ++i) {
        while (wall_clock64() - t0 < (i + 1) * step) __builtin_amdgcn_s_sleep(10);
        __hip_atomic_store(sig[order[i]], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall_clock64() - t0 < (nsig + 2) * step) __builtin_amdgcn_s_sleep(10);
}

// fires sig[i] at times[i] ticks after start (times ascending)
__global__ void timed_producer(long long **sig, const long long *times, int nsig, long long tail) {
    long long t0 = wall_clock64();
    for (int i = 0; i < nsig; ++i) {
        while (wall_clock64() - t0 < times[i]) __builtin_amdgcn_s_sleep(2);
        __hip_atomic_store(sig[i], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall_clock64() - t0 < tail) __builtin_amdgcn_s_sleep(10);
}

int main() {
    CK(hipSetDevice(0));
    HK(hsa_init());
    HK(hsa_iterate_agents(cb, nullptr));
    printf("agents: %zu gpu, %zu cpu\n", gpus.size(), cpus.size());
    const int n = 1 << 20;                                    // 8 MB
    double *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, 8 * n)); CK(hipHostMalloc(&h, 8 * n, hipHostMallocDefault));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
        hsa_signal_t dep, done;
        HK(hsa_signal_create(1, 0, nullptr, &dep)); HK(hsa_signal_create(1, 0, nullptr, &done));
        long long *depval = const_cast<long long *>(reinterpret_cast<volatile long long *>(&reinterpret_cast<amd_signal_t *>(dep.handle)->value));
        memset(h, 0, 8 * n);
        const double tag = 100.0 * mode + rep + 1;
        CK(hipMemsetAsync(d, 0, 8 * n, s)); CK(hipStreamSynchronize(s));
        double t0 = now();
        // order A (rep 0,1): copy submitted BEFORE the kernel is launched; order B (rep 2): after
        if (rep < 2) HK(hsa_amd_memory_async_copy(h, cpus[0], d, gpus[0], 8 * n, 1, &dep, done));
        hipLaunchKernelGGL(producer, dim3(1), dim3(256), 0, s, d, n, tag, 500000LL /* 5 ms */, 500000LL, depval, mode);
        if (rep == 2) HK(hsa_amd_memory_async_copy(h, cpus[0], d, gpus[0], 8 * n, 1, &dep, done));
        double t1 = now();
        hsa_signal_value_t v = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 4000000000ull, HSA_WAIT_STATE_BLOCKED);
        double t2 = now();
        CK(hipStreamSynchronize(s));
        double t3 = now();
        int bad = 0; for (int i = 0; i < n; i += 997) if (h[i] != tag) ++bad;
        printf("mode %d rep %d: submit %.3f ms; copy complete at %.3f ms (signal value %ld); kernel complete at %.3f ms; wrong values %d (h[0] = %.1f)\n", mode, rep,
               (t1 - t0) * 1e3, (t2 - t0) * 1e3, (long)v, (t3 - t0) * 1e3, bad, h[0]);
        hsa_signal_destroy(dep); hsa_signal_destroy(done);
    }
    // several copies in flight, dependencies fired out of submission order; reused (re-armed) signals on the second pass
    {
        const int K = 6;
        hsa_signal_t dep[K], done[K];
        long long **dsig = nullptr; int *dorder = nullptr;
        CK(hipHostMalloc(&dsig, K * sizeof(long long *), hipHostMallocDefault)); CK(hipHostMalloc(&dorder, K * sizeof(int), hipHostMallocDefault));
        for (int i = 0; i < K; ++i) { HK(hsa_signal_create(1, 0, nullptr, &dep[i])); HK(hsa_signal_create(1, 0, nullptr, &done[i]));
            dsig[i] = const_cast<long long *>(reinterpret_cast<volatile long long *>(&reinterpret_cast<amd_signal_t *>(dep[i].handle)->value)); }
        const int orders[2][K] = {{2, 0, 1, 3, 5, 4}, {5, 4, 3, 2, 1, 0}};
        for (int pass = 0; pass < 3; ++pass) {
            const int *ord = orders[pass % 2];
            for (int i = 0; i < K; ++i) { dorder[i] = ord[i]; hsa_signal_store_relaxed(dep[i], 1); hsa_signal_store_relaxed(done[i], 1); }
            double t0 = now();
            hipLaunchKernelGGL(multi_producer, dim3(1), dim3(1), 0, s, dsig, dorder, K, 100000LL /* 1 ms */);
            for (int i = 0; i < K; ++i) HK(hsa_amd_memory_async_copy(h + (size_t)i * (n / K), cpus[0], d + (size_t)i * (n / K), gpus[0], 8 * (size_t)(n / K), 1, &dep[i], done[i]));
            double tdone[K]; bool got[K] = {false};
            int left = K;
            while (left && now() - t0 < 2.0) {
                for (int i = 0; i < K; ++i) if (!got[i] && hsa_signal_load_relaxed(done[i]) <= 0) { got[i] = true; tdone[i] = now() - t0; --left; }
            }
            CK(hipStreamSynchronize(s));
            printf("pass %d: dependency of copy i fired at (ms):", pass);
            for (int i = 0; i < K; ++i) { int when = 0; for (int q = 0; q < K; ++q) if (ord[q] == i) when = q + 1; printf(" c%d@%d", i, when); }
            printf("  | copies completed at:");
            for (int i = 0; i < K; ++i) printf(" c%d %.2f", i, got[i] ? tdone[i] * 1e3 : -1.0);
            printf("\n");
        }
    }
    // replica of the application's pattern: 11 transfers per iteration [A 16.8 MB, q|l|u 40 KB, r 8 B, g0..g7 ~8 MB each], re-armed signals,
    // the producer fires A/q/r at 0.46 ms, g0-g3 at 0.60 ms, g4-g6 at 1.25 ms, g7 at 1.29 ms
    {
        const int K = 11;
        const size_t bytes[K] = {16777216, 40960, 8, 7900000, 8800000, 9400000, 5500000, 9200000, 10400000, 7600000, 8100000};
        const int fire_at_us[K] = {460, 460, 460, 600, 600, 600, 600, 1250, 1250, 1250, 1290};
        double *d2 = nullptr, *h2 = nullptr;
        size_t tot = 0; for (int i = 0; i < K; ++i) tot += (bytes[i] + 255) / 256 * 256;
        CK(hipMalloc(&d2, tot)); CK(hipHostMalloc(&h2, tot, hipHostMallocDefault));
        hsa_signal_t dep[K], done[K];
        long long **dsig = nullptr; int *dorder = nullptr; long long *dtimes = nullptr;
        CK(hipHostMalloc(&dsig, K * sizeof(long long *), hipHostMallocDefault)); CK(hipHostMalloc(&dtimes, K * sizeof(long long), hipHostMallocDefault));
        for (int i = 0; i < K; ++i) { HK(hsa_signal_create(1, 0, nullptr, &dep[i])); HK(hsa_signal_create(1, 0, nullptr, &done[i]));
            dsig[i] = const_cast<long long *>(reinterpret_cast<volatile long long *>(&reinterpret_cast<amd_signal_t *>(dep[i].handle)->value)); dtimes[i] = fire_at_us[i] * 100LL; }
        for (int it = 0; it < 4; ++it) {
            for (int i = 0; i < K; ++i) { hsa_signal_store_relaxed(dep[i], 1); hsa_signal_store_relaxed(done[i], 1); }
            double t0 = now();
            hipLaunchKernelGGL(timed_producer, dim3(1), dim3(1), 0, s, dsig, dtimes, K, 150000LL);
            size_t off = 0;
            for (int i = 0; i < K; ++i) { HK(hsa_amd_memory_async_copy((char *)h2 + off, cpus[0], (char *)d2 + off, gpus[0], bytes[i], 1, &dep[i], done[i])); off += (bytes[i] + 255) / 256 * 256; }
            double tdone[K]; bool got[K] = {false}; int left = K;
            while (left && now() - t0 < 2.0)
                for (int i = 0; i < K; ++i) if (!got[i] && hsa_signal_load_relaxed(done[i]) <= 0) { got[i] = true; tdone[i] = now() - t0; --left; }
            CK(hipStreamSynchronize(s));
            printf("iteration %d: completed at (us):", it);
            for (int i = 0; i < K; ++i) printf(" %s%d %.0f", i < 3 ? "x" : "g", i < 3 ? i : i - 3, got[i] ? tdone[i] * 1e6 : -1.0);
            printf("\n");
        }
    }
    return 0;
}
