// How fast do 512 workgroups x 8 waves write an 80 MB row-major term array as contiguous RUNS of `run` bytes (a row's share of a column band),
// rows `row_bytes` apart — the write phase of sparse_block_kernel without any of its LDS work?  Block (rb, cb) writes rows rb*RB .. +RB-1,
// each run at row * row_bytes + cb * run.   build: hipcc --offload-arch=gfx950 -O2 tools/run_store_probe.hip -o tools/run_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void run_stores(char *out, int nrb, int rb_rows, long long row_bytes, int run) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    const int cb = (b / (8 * nrb)) * 8 + (b & 7), rb = (b >> 3) % nrb;
    const int rows_per_wave = rb_rows / 8;
    for (int i = 0; i < rows_per_wave; ++i) {
        const long long row = (long long)rb * rb_rows + wave * rows_per_wave + i;
        char *seg = out + row * row_bytes + (long long)cb * run;
        u64x2 v; v.x = (u64)row; v.y = (u64)lane;
        for (int c = lane; c * 16 < run; c += 64) *reinterpret_cast<u64x2 *>(seg + c * 16) = v;
    }
}

int main() {
    const long long rows = 4096, row_bytes = 19648;            // 818.7 terms x 24 B, rounded to 16
    char *d = nullptr;
    CK(hipMalloc(&d, rows * row_bytes + (1 << 20)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int rb_rows, run; const char *what; } cfgs[] = {
        {128, 1228, "128-row blocks x 16 bands: runs of 1228 B (shipped geometry)"},
        {64, 2456, "64-row blocks x 8 bands: runs of 2456 B"},
        {256, 614, "256-row blocks x 32 bands: runs of 614 B"},
        {32, 4912, "32-row blocks x 4 bands: runs of 4912 B"},
    };
    for (auto &c : cfgs) {
        const int nrb = (int)(rows / c.rb_rows), ncb = (int)(row_bytes / c.run);
        const int run16 = c.run / 16 * 16;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(run_stores, dim3(nrb * ncb), dim3(512), 0, s, d, nrb, c.rb_rows, row_bytes, run16);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)rows * ncb * run16;
            if (rep) printf("%-66s %d workgroups  %.1f us  %.2f TB/s (%.0f MB)\n", c.what, nrb * ncb, ms * 1e3, bytes / ms / 1e9, bytes / 1e6);
        }
    }
    return 0;
}
