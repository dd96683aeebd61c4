// Microbenchmark: do the f64 matrix pipe (v_mfma_f64_4x4x4_4b_f64) and the f64 vector FMA (v_fma_f64) of one SIMD run
// concurrently on gfx950, or do they share the double-precision datapath?
// One 512-thread workgroup per CU: waves w and w + 4 share a SIMD.  Roles per wave: 0 = idle, 1 = MFMA stream, 2 = v_fma_f64 stream,
// 3 = one wave interleaving both (1 MFMA : 4 FMA = equal FLOP).
//   T(mfma alone), T(valu alone), T(mfma beside valu on the same SIMD): if the last is ~max of the first two the pipes are separate
//   (and a Gram kernel could feed both); if it is ~their sum they share the FP64 units.
// Build: hipcc --offload-arch=gfx950 -O3 tools/f64_coissue.hip -o tools/f64_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512, 2) void k(double *out, int iters, int role_lo, int role_hi) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, b, acc[i]);
        }
    } else if (role == 3) {
        double vacc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) vacc[i] = 0.0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) vacc[(4 * i + r) & 15] = __builtin_fma(a, b, vacc[(4 * i + r) & 15]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += vacc[i];
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(double *out, int iters, int lo, int hi) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 200, lo, hi);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, lo, hi);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    double *out; (void)hipMalloc(&out, 256 * 512 * 8);
    const int iters = 20000;
    // per wave: role 1 = iters*16 MFMAs * 512 FLOP; role 2 = iters*64 FMAs * 128 FLOP; role 3 = both
    const double wave_flop = (double)iters * 16 * 512;
    struct { const char *name; int lo, hi; double waves_mfma, waves_valu; } cases[] = {
        {"MFMA on 1 wave/SIMD, partner idle", 1, 0, 4, 0},
        {"VALU on 1 wave/SIMD, partner idle", 2, 0, 0, 4},
        {"MFMA on both waves of a SIMD", 1, 1, 8, 0},
        {"VALU on both waves of a SIMD", 2, 2, 0, 8},
        {"MFMA wave + VALU wave per SIMD", 1, 2, 4, 4},
        {"interleaved in one wave, partner idle", 3, 0, 4, 4},
        {"interleaved in both waves", 3, 3, 8, 8},
    };
    for (auto &c : cases) {
        float ms = run(out, iters, c.lo, c.hi);
        double tf_m = c.waves_mfma * 256 * wave_flop / ms / 1e9, tf_v = c.waves_valu * 256 * wave_flop / ms / 1e9;
        printf("%-40s %8.3f ms   MFMA %6.2f + VALU %6.2f = %6.2f TFLOP/s\n", c.name, ms, tf_m, tf_v, tf_m + tf_v);
    }
    return 0;
}
