// Does the hardware honour cbsz/abid (A-block broadcast) on v_mfma_f64_4x4x4_4b_f64?  (hipcc's builtin drops the modifiers.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define MF(CB, AB) \
    asm volatile("s_nop 7\n v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0 cbsz:" #CB " abid:" #AB "\n s_nop 15\n s_nop 15\n" : "=v"(d) : "v"(a), "v"(b))
template <int V>
__global__ void probe(double *out) {
    int lane = threadIdx.x;
    for (int t = 0; t < 64; ++t) {
        double a = ldexp(1.0, lane);
        double b = (lane == t) ? 1.0 : 0.0;
        double d;
        if (V == 0) MF(2, 0); else if (V == 1) MF(2, 1); else if (V == 2) MF(2, 3); else MF(1, 1);
        out[t * 64 + lane] = d;
    }
}
template <int V>
void run(double *dout, const char *name) {
    hipLaunchKernelGGL((probe<V>), dim3(1), dim3(64), 0, 0, dout);
    static double h[64 * 64];
    (void)hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    printf("== %s\n", name);
    for (int t = 0; t < 64; t += 1) {
        if (!(t < 8 || (t >= 16 && t < 24))) continue;
        printf("t=%2d:", t);
        for (int l = 0; l < 64; ++l) if (h[t * 64 + l] != 0.0) printf(" %d<-%d", l, (int)log2(h[t * 64 + l]));
        printf("\n");
    }
}
int main() {
    double *dout; (void)hipMalloc(&dout, 64 * 64 * 8);
    run<0>(dout, "cbsz:2 abid:0"); run<1>(dout, "cbsz:2 abid:1"); run<2>(dout, "cbsz:2 abid:3"); run<3>(dout, "cbsz:1 abid:1");
    return 0;
}
