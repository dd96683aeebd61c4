// Does the nontemporal policy raise the chip's streaming store / copy rate?  Plain vs __builtin_nontemporal_store / _load, 16 bytes per lane,
// 2048 workgroups x 256 threads, grid-stride.  build: hipcc --offload-arch=gfx950 -O2 tools/nt_store_probe.hip -o tools/nt_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int NT> __global__ __launch_bounds__(256) void fill(f64x2 *out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    f64x2 v; v.x = (double)threadIdx.x; v.y = 1.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <int NT> __global__ __launch_bounds__(256) void copy(const f64x2 *in, f64x2 *out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const f64x2 v = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT & 2) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <int NT> __global__ __launch_bounds__(256) void readsum(const f64x2 *in, double *sink, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const f64x2 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long long mb : {80ll, 200ll, 1000ll}) {
        const long long n = mb * 1000000 / 16;
        f64x2 *a, *b; double *sink;
        CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&sink, 8));
        CK(hipMemset(a, 0, n * 16));
        auto time = [&](const char *what, auto launch, double bytes) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0, s); launch(); hipEventRecord(e1, s); hipStreamSynchronize(s);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
            }
            printf("%5lld MB  %-28s %.1f us  %.2f TB/s\n", mb, what, best * 1e3, bytes / best / 1e9);
        };
        const dim3 g(2048), t(256);
        time("fill plain", [&] { hipLaunchKernelGGL(fill<0>, g, t, 0, s, a, n); }, n * 16.0);
        time("fill nt", [&] { hipLaunchKernelGGL(fill<1>, g, t, 0, s, a, n); }, n * 16.0);
        time("read plain", [&] { hipLaunchKernelGGL(readsum<0>, g, t, 0, s, a, sink, n); }, n * 16.0);
        time("read nt", [&] { hipLaunchKernelGGL(readsum<1>, g, t, 0, s, a, sink, n); }, n * 16.0);
        time("copy plain", [&] { hipLaunchKernelGGL(copy<0>, g, t, 0, s, a, b, n); }, n * 32.0);
        time("copy nt loads", [&] { hipLaunchKernelGGL(copy<1>, g, t, 0, s, a, b, n); }, n * 32.0);
        time("copy nt stores", [&] { hipLaunchKernelGGL(copy<2>, g, t, 0, s, a, b, n); }, n * 32.0);
        time("copy nt both", [&] { hipLaunchKernelGGL(copy<3>, g, t, 0, s, a, b, n); }, n * 32.0);
        hipFree(a); hipFree(b); hipFree(sink);
    }
    return 0;
}
