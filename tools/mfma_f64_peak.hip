// Microbenchmark: sustained fp64 rates on MI355X (the guides list no f64 MFMA peak).
//   - v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 with independent accumulators
//   - v_fma_f64 (VALU) with independent accumulators
// Reports TFLOP/s from HIP events and shader cycles per instruction per wave from s_memtime (clock64), which is
// independent of DVFS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else if (MODE == 1) acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
            else {
                acc[i][0] = __builtin_fma(a, b, acc[i][0]); acc[i][1] = __builtin_fma(a, b, acc[i][1]);
                acc[i][2] = __builtin_fma(a, b, acc[i][2]); acc[i][3] = __builtin_fma(a, b, acc[i][3]);
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, double flop_per_inst, int inst_per_iter, double *out, long long *cyc) {
    for (int wps = 1; wps <= 2; ++wps) {
        int blocks = 256 * wps, iters = 4000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double insts = (double)blocks * 4 * iters * inst_per_iter;
        printf("%-28s waves/SIMD=%d  %8.3f ms  %7.2f TFLOP/s   %6.1f memtime-ticks per instruction per wave\n", name, wps, ms,
               insts * flop_per_inst / ms / 1e9, (double)c / ((double)iters * inst_per_iter));
    }
}
int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 4096 * 512 * 8); (void)hipMalloc(&cyc, 8);
    run<0>("v_mfma_f64_16x16x4_f64", 2.0 * 16 * 16 * 4, 16, out, cyc);
    run<1>("v_mfma_f64_4x4x4_4b_f64", 2.0 * 4 * 4 * 4 * 4, 16, out, cyc);
    run<2>("v_fma_f64 (VALU)", 2.0 * 64, 64, out, cyc);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; (void)hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("clockRate=%d kHz wallClockRate=%d kHz\n", clk, wclk);
    return 0;
}
