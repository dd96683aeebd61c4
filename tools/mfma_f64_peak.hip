// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (the guides list no f64 MFMA peak).
// Every wave runs ITER x 16 independent-accumulator MFMAs; reports TFLOP/s for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out; hipMalloc(&out, 4096 * 512 * 8);
    for (int wps = 1; wps <= 2; ++wps) {
        int blocks = 256 * wps, iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 4;
        printf("waves/SIMD=%d  %.3f ms  %.2f TFLOP/s f64 MFMA  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wps, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
    return 0;
}
