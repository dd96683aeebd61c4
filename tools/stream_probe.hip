// What does the memory system give a kernel that streams a tall column-major panel (rows >> columns) once, by access pattern?
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe [rows] [cols]
// Every kernel reads all rows x cols doubles once, adds them up per lane (so that no load is dead) and writes one double per wave.
//   flat      the copy pattern: 1 KB contiguous per wave instruction, consecutive waves consecutive KBs of a column, D loads in flight
//   colwave   a wave iteration = 128 rows of every column: one 1 KB load per column (what a wave-private transpose would issue)
//   mfma      the MFMA operand layout read straight from global memory (gram_stream_kernel): lane (lm, lk) loads 16 bytes of column lm at rows
//             8 i + 2 lk: 64 contiguous bytes of each of 16 columns per instruction, IT instructions per iteration, D iterations in flight
//   panel     the panel kernels' pattern (gram_narrow_kernel): a workgroup loads a stage of R rows of every column, LPC lanes walking down a column
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int D>
__global__ __launch_bounds__(256) void flat_kernel(const double *A, long lda, long rows, int cols, double *out) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long per_col = rows / 128, units = per_col * cols;          // 1 KB units
    double acc = 0.0;
    for (long u0 = gw; u0 < units; u0 += W * D) {
        f64x2 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            long u = u0 + d * W; if (u >= units) u = u0;
            v[d] = *reinterpret_cast<const f64x2 *>(A + (u / per_col) * lda + (u % per_col) * 128 + 2 * lane);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d].x + v[d].y;
    }
    if (lane == 0) out[gw] = acc;
    else if (acc == 1.2345e-300) out[gw] = acc;
}

template <int NC, int D>
__global__ __launch_bounds__(256) void colwave_kernel(const double *A, long lda, long rows, int cols, double *out) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nit = rows / 128;
    double acc = 0.0;
    for (long it = gw; it < nit; it += W) {
        for (int c0 = 0; c0 < NC; c0 += D) {
            f64x2 v[D];
#pragma unroll
            for (int d = 0; d < D; ++d) v[d] = *reinterpret_cast<const f64x2 *>(A + (long)(c0 + d) * lda + it * 128 + 2 * lane);
#pragma unroll
            for (int d = 0; d < D; ++d) acc += v[d].x + v[d].y;
        }
    }
    if (lane == 0) out[gw] = acc;
    else if (acc == 1.2345e-300) out[gw] = acc;
}

template <int NB, int IT, int D>
__global__ __launch_bounds__(256) void mfma_kernel(const double *A, long lda, long rows, int cols, double *out) {
    const int lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
    const long W = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nit = rows / (8 * IT);
    const int my = (int)(nit > gw ? (nit - gw + W - 1) / W : 0);
    double acc = 0.0;
    f64x2 buf[D][NB][IT];
    auto load = [&](int d, int s) {
        const long row0 = (gw + (long)min(s, my - 1) * W) * 8 * IT;
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int i = 0; i < IT; ++i) buf[d][t][i] = *reinterpret_cast<const f64x2 *>(A + (long)(16 * t + lm) * lda + row0 + 8 * i + 2 * lk);
    };
    if (my > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load(d, d);
        for (int s0 = 0; s0 < my; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (s0 + d < my) {
#pragma unroll
                    for (int t = 0; t < NB; ++t)
#pragma unroll
                        for (int i = 0; i < IT; ++i) acc += buf[d][t][i].x + buf[d][t][i].y;
                }
                load(d, s0 + d + D);
            }
        }
    }
    if (lane == 0) out[gw] = acc;
    else if (acc == 1.2345e-300) out[gw] = acc;
}

// a workgroup's stage: R rows of NC columns, thread (kp = tid % LPC, cc = tid / LPC) loads row pairs 2 kp + 2 LPC j of columns cc + (256 / LPC) q
template <int NC, int R, int LPC>
__global__ __launch_bounds__(256) void panel_kernel(const double *A, long lda, long rows, int cols, double *out) {
    constexpr int NCC = 256 / LPC, NQ = NC > NCC ? NC / NCC : 1, NJ = R / (2 * LPC);
    const int tid = threadIdx.x, kp = tid % LPC, cc = tid / LPC;
    const long G = gridDim.x, nst = rows / R;
    double acc = 0.0;
    for (long st = blockIdx.x; st < nst; st += G) {
        f64x2 v[NQ][NJ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = cc + NCC * q;
                v[q][j] = col < NC ? *reinterpret_cast<const f64x2 *>(A + (long)col * lda + st * R + 2 * LPC * j + 2 * kp) : f64x2{0.0, 0.0};
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc += v[q][j].x + v[q][j].y;
    }
    if ((tid & 63) == 0) out[blockIdx.x * 4 + (tid >> 6)] = acc;
    else if (acc == 1.2345e-300) out[0] = acc;
}

static double *A, *out;
static long lda, rows;
static int cols;
static hipEvent_t e0, e1;

template <typename F>
static void timeit(const char *label, int G, F launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = 8.0 * rows * cols;
    printf("  %-34s G=%5d  %8.1f us  %6.2f TB/s  (%s)\n", label, G, us, bytes / us / 1e6, hipGetErrorString(hipGetLastError()));
}

#define RUN(label, G, kernel) timeit(label, G, [&] { hipLaunchKernelGGL((kernel), dim3(G), dim3(256), 0, 0, A, lda, rows, cols, out); })

int main(int argc, char **argv) {
    rows = argc > 1 ? atol(argv[1]) : 1L << 20;
    cols = argc > 2 ? atoi(argv[2]) : 16;
    lda = rows + 64;
    hipMalloc(&A, sizeof(double) * lda * cols);
    hipMalloc(&out, sizeof(double) * 65536);
    hipMemset(A, 0, sizeof(double) * lda * cols);
    hipEventCreate(&e0); hipEventCreate(&e1);
    // spin the clocks up
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((flat_kernel<4>), dim3(1024), dim3(256), 0, 0, A, lda, rows, cols, out);
    hipDeviceSynchronize();
    printf("%ld x %d doubles (%.1f MB), lda %ld\n", rows, cols, 8.0 * rows * cols / 1e6, lda);
    for (int G : {256, 512, 1024, 2048, 4096}) {
        RUN("flat D=1", G, flat_kernel<1>);
        RUN("flat D=4", G, flat_kernel<4>);
        RUN("flat D=8", G, flat_kernel<8>);
    }
    if (cols == 16) {
        for (int G : {256, 512, 1024, 2048}) {
            RUN("colwave 16 cols D=4", G, (colwave_kernel<16, 4>));
            RUN("colwave 16 cols D=16", G, (colwave_kernel<16, 16>));
            RUN("mfma NB=1 IT=4 D=1", G, (mfma_kernel<1, 4, 1>));
            RUN("mfma NB=1 IT=4 D=2", G, (mfma_kernel<1, 4, 2>));
            RUN("mfma NB=1 IT=4 D=4", G, (mfma_kernel<1, 4, 4>));
            RUN("mfma NB=1 IT=8 D=2", G, (mfma_kernel<1, 8, 2>));
            RUN("mfma NB=1 IT=16 D=1", G, (mfma_kernel<1, 16, 1>));
            RUN("mfma NB=1 IT=16 D=2", G, (mfma_kernel<1, 16, 2>));
            RUN("panel 16 cols R=256 LPC=16", G, (panel_kernel<16, 256, 16>));
            RUN("panel 16 cols R=512 LPC=16", G, (panel_kernel<16, 512, 16>));
        }
    } else if (cols == 64) {
        for (int G : {256, 512, 1024, 2048}) {
            RUN("colwave 64 cols D=8", G, (colwave_kernel<64, 8>));
            RUN("colwave 64 cols D=16", G, (colwave_kernel<64, 16>));
            RUN("mfma NB=4 IT=2 D=2", G, (mfma_kernel<4, 2, 2>));
            RUN("mfma NB=4 IT=4 D=1", G, (mfma_kernel<4, 4, 1>));
            RUN("mfma NB=4 IT=4 D=2", G, (mfma_kernel<4, 4, 2>));
            RUN("mfma NB=4 IT=8 D=1", G, (mfma_kernel<4, 8, 1>));
            RUN("panel 64 cols R=64 LPC=8", G, (panel_kernel<64, 64, 8>));
            RUN("panel 64 cols R=128 LPC=8", G, (panel_kernel<64, 128, 8>));
        }
    }
    return 0;
}
