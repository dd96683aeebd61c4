// How fast can 256 persistent 8-wave workgroups (the contraction's grid) write tile epilogues?  Each workgroup writes `tiles` tiles of 128 rows x
// 3072 bytes (128 QuadraticTerms), rows `row_stride` bytes apart, as 16-byte stores — the access pattern of sk_epilogue without any of its
// arithmetic or LDS traffic.  Compared with the same bytes from 2048 workgroups.   build: hipcc --offload-arch=gfx950 -O2 tools/store_probe.hip -o tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void epilogue_stores(u64 *out, long long tile_bytes, int tiles, long long row_stride, int rows_per_pass) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = 0; t < tiles; ++t) {
        char *base = reinterpret_cast<char *>(out) + ((long long)(blockIdx.x * tiles + t)) * tile_bytes;
        for (int row = wave; row < 128; row += 8) {
            char *seg = base + (long long)row * row_stride;
            u64x2 v; v.x = (u64)row; v.y = (u64)lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<u64x2 *>(seg + (c * 64 + lane) * 16) = v;
        }
        if (rows_per_pass) __syncthreads();
    }
}

int main() {
    const long long total = 201ll << 20;                     // ~ the 201 MB of one launch
    u64 *d = nullptr;
    CK(hipMalloc(&d, total + (1 << 20)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int blocks, tiles; long long stride; const char *what; } cfgs[] = {
        {256, 2, 3072, "256 workgroups x 2 tiles, rows contiguous (3 KB apart)"},
        {256, 2, 98304, "256 workgroups x 2 tiles, rows 96 KB apart (row-major triangle)"},
        {512, 1, 3072, "512 workgroups x 1 tile, rows contiguous"},
        {512, 1, 98304, "512 workgroups x 1 tile, rows 96 KB apart"},
    };
    for (auto &c : cfgs) {
        const long long tile_bytes = c.stride == 3072 ? 128 * 3072 : 3072;   // strided: tiles interleave like the columns of a tile row
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(epilogue_stores, dim3(c.blocks), dim3(512), 0, s, d, tile_bytes, c.tiles, c.stride, 1);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)c.blocks * c.tiles * 128 * 3072;
            if (rep) printf("%-66s %.1f us  %.2f TB/s (%.0f MB)\n", c.what, ms * 1e3, bytes / ms / 1e9, bytes / 1e6);
        }
    }
    return 0;
}
