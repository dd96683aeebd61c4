// Probe for the host-delivery design: can a copy stream be made to wait for IN-KERNEL progress?
//   1. hipDeviceAttributeCanUseStreamWaitValue; hipStreamWaitValue32(GTE) on a counter a running kernel increments (device memory,
//      signal memory, host-mapped memory), followed by a D2H copy: does the copy start while the kernel is still running?
//   2. page-locked D2H rate for 8 / 17 / 67 MB in one piece and in 4 MB pieces; two copies on two streams at once.
//   3. kernel stores straight into host-mapped memory (zero copy): rate.
// build: hipcc --offload-arch=gfx950 -O2 tools/deliver_probe.hip -o tools/deliver_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// every `step_cycles` the single workgroup bumps *counter (system-scope release) until it reaches `n`; fills data[i] = i+1 first
__global__ void progress_kernel(unsigned *counter, int n, long long step_cycles, double *data, int per_step) {
    for (int s = 0; s < n; ++s) {
        for (int i = threadIdx.x; i < per_step; i += blockDim.x) data[(size_t)s * per_step + i] = (double)(s + 1);
        __syncthreads();
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < step_cycles) __builtin_amdgcn_s_sleep(10);
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
    }
}

__global__ void store_kernel(double *dst, size_t n) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    d2 v; v.x = 1.0; v.y = 2.0;
    for (size_t k = i; k < n / 2; k += stride) reinterpret_cast<d2 *>(dst)[k] = v;
}

int main() {
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t s, c, c2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c2, hipStreamNonBlocking));
    const int steps = 8, per_step = 1 << 20;                       // 8 MB per step
    double *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, sizeof(double) * steps * per_step));
    CK(hipHostMalloc(&h, sizeof(double) * steps * per_step, hipHostMallocDefault));
    memset(h, 0, sizeof(double) * steps * per_step);
    if (can) {
        for (int kind = 0; kind < 3; ++kind) {
            unsigned *ctr = nullptr;
            const char *name = kind == 0 ? "hipMalloc" : kind == 1 ? "signal memory" : "host-mapped";
            hipError_t e = kind == 0 ? hipMalloc(&ctr, 64) : kind == 1 ? hipExtMallocWithFlags((void **)&ctr, 64, hipMallocSignalMemory)
                                                                       : hipHostMalloc(&ctr, 64, hipHostMallocMapped);
            if (e != hipSuccess) { printf("%s: allocation failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 64, s));
                CK(hipStreamSynchronize(s));
                memset(h, 0, sizeof(double) * steps * per_step);
                hipEvent_t k0, k1, ce[steps];
                CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
                for (int i = 0; i < steps; ++i) CK(hipEventCreate(&ce[i]));
                CK(hipEventRecord(k0, s));
                hipLaunchKernelGGL(progress_kernel, dim3(1), dim3(256), 0, s, ctr, steps, 20000000LL /* 100 MHz wall clock: 200 ms? see print */ / 100, d, per_step);
                CK(hipEventRecord(k1, s));
                bool ok = true;
                for (int i = 0; i < steps && ok; ++i) {
                    hipError_t w = hipStreamWaitValue32(c, ctr, (unsigned)(i + 1), hipStreamWaitValueGte, 0xffffffffu);
                    if (w != hipSuccess) { printf("%s: hipStreamWaitValue32 -> %s\n", name, hipGetErrorString(w)); (void)hipGetLastError(); ok = false; break; }
                    CK(hipMemcpyAsync(h + (size_t)i * per_step, d + (size_t)i * per_step, sizeof(double) * per_step, hipMemcpyDeviceToHost, c));
                    CK(hipEventRecord(ce[i], c));
                }
                if (!ok) { CK(hipStreamSynchronize(s)); break; }
                CK(hipStreamSynchronize(c));
                CK(hipStreamSynchronize(s));
                float kms = 0; CK(hipEventElapsedTime(&kms, k0, k1));
                printf("%s rep %d: kernel %.3f ms;", name, rep, kms);
                int bad = 0;
                for (int i = 0; i < steps; ++i) {
                    float t = 0; CK(hipEventElapsedTime(&t, k0, ce[i]));
                    printf(" copy%d done @%.3f", i, t);
                    for (int q = 0; q < per_step; q += 4097) if (h[(size_t)i * per_step + q] != (double)(i + 1)) ++bad;
                }
                printf("; wrong values %d\n", bad);
            }
            if (kind == 2) (void)hipHostFree(ctr); else (void)hipFree(ctr);
        }
    }
    // 2. D2H rates
    const size_t sizes[] = {8u << 20, 17u << 20, 67u << 20};
    for (size_t sz : sizes) {
        if (sz > sizeof(double) * steps * per_step) continue;
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, c));
            CK(hipStreamSynchronize(c));
            double t1 = now();
            for (size_t o = 0; o < sz; o += 4u << 20) CK(hipMemcpyAsync((char *)h + o, (char *)d + o, std::min<size_t>(4u << 20, sz - o), hipMemcpyDeviceToHost, c));
            CK(hipStreamSynchronize(c));
            double t2 = now();
            // two halves on two streams
            CK(hipMemcpyAsync(h, d, sz / 2, hipMemcpyDeviceToHost, c));
            CK(hipMemcpyAsync((char *)h + sz / 2, (char *)d + sz / 2, sz - sz / 2, hipMemcpyDeviceToHost, c2));
            CK(hipStreamSynchronize(c)); CK(hipStreamSynchronize(c2));
            double t3 = now();
            printf("D2H %zu MB: one piece %.3f ms (%.1f GB/s); 4 MB pieces %.3f ms (%.1f GB/s); two streams %.3f ms (%.1f GB/s)\n", sz >> 20, (t1 - t0) * 1e3,
                   sz / (t1 - t0) / 1e9, (t2 - t1) * 1e3, sz / (t2 - t1) / 1e9, (t3 - t2) * 1e3, sz / (t3 - t2) / 1e9);
        }
    }
    // 3. zero-copy stores
    double *hm = nullptr, *hd = nullptr;
    CK(hipHostMalloc(&hm, 64u << 20, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hd, hm, 0));
    for (int cfg = 0; cfg < 7; ++cfg) {
        const int blocks = (const int[]){64, 256, 1024, 8, 8, 16, 4}[cfg], threads = (const int[]){256, 256, 256, 1024, 256, 1024, 1024}[cfg];
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(threads), 0, s, hd, (size_t)(64u << 20) / 8);
            CK(hipStreamSynchronize(s));
            double t1 = now();
            printf("zero-copy store 64 MB, %d workgroups x %d threads: %.3f ms (%.1f GB/s) host sees %.1f %.1f\n", blocks, threads, (t1 - t0) * 1e3, (64u << 20) / (t1 - t0) / 1e9, hm[0], hm[1]);
        }
    }
    return 0;
}
