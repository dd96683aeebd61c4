// Empirical lane-layout probe for v_mfma_f64_4x4x4_4b_f64 (and its cbsz/abid broadcast) on gfx950.
// For each one-hot B lane t (B = 1.0 in lane t only) and A = 2^lane, D[out lane] = 2^(A lane paired with t).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CBSZ, int ABID>
__global__ void probe(double *out) {
    int lane = threadIdx.x;
    for (int t = 0; t < 64; ++t) {
        double a = ldexp(1.0, lane);
        double b = (lane == t) ? 1.0 : 0.0;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
        out[t * 64 + lane] = d;
    }
}
template <int CBSZ, int ABID>
void run(double *dout) {
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dout);
    static double h[64 * 64];
    (void)hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    printf("== cbsz=%d abid=%d : for B one-hot lane t -> list of (out_lane <- A_lane)\n", CBSZ, ABID);
    for (int t = 0; t < 64; ++t) {
        printf("t=%2d:", t);
        for (int l = 0; l < 64; ++l) if (h[t * 64 + l] != 0.0) printf(" %d<-%d", l, (int)log2(h[t * 64 + l]));
        printf("\n");
    }
}
int main() {
    double *dout; (void)hipMalloc(&dout, 64 * 64 * 8);
    run<0, 0>(dout);
    run<2, 0>(dout);
    run<2, 1>(dout);
    run<2, 3>(dout);
    return 0;
}
