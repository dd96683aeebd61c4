// Batched small instances (BASELINE config 4: n = r = 128): ONE workgroup per instance, the whole least-squares objective of an
// instance — Q = 2 A'A (upper triangle), q = 2 A'c, c'c — from a single pass over its A.
//
// The general batch path (gram_sk.hip: one workgroup per (instance, 128x128 tile), 16-row stages with a barrier each, plus a second
// kernel that re-reads A for q) spends 1.1 ms on 8192 instances; the data is 1.9 GB, i.e. ~0.3 ms of HBM time.  Here
//   * A is streamed ONCE, 64 rows at a time, into LDS (K-contiguous columns, odd pitch — the operand layout of gram_sk.hip); the
//     chunk is 66 KB, so TWO workgroups share a CU and one loads while the other multiplies;
//   * only the 36 of the 64 16x16 sub-tiles that touch the upper triangle are computed.  Sub-tiles are dealt to the 8 waves as
//     "four of one column strip + one of another" so that the four rotated B-operand reads of a strip are shared (5 + 8 LDS reads per
//     20 MFMAs) and every SIMD carries 9 sub-tiles (the slots a role does not need are compiled out per role class);
//   * q is accumulated from the same LDS chunk by the vector ALU and the coefficients leave through an LDS transposition as
//     contiguous row segments; c'c (a serial left-to-right chain per instance, src/functions.jl:574) stays in its own kernel, one
//     thread per instance — inside this kernel the chain stalls a whole workgroup.
// Coefficient order within a dot product differs from the general path only in q (tolerance 1e-12); Q uses the same MFMA lane
// mapping and k order as gram_sk.hip.
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef PMT_BS_SKIP
#define PMT_BS_SKIP 0      // profiling only: 1 = no contraction, 2 = no q / c'c, 4 = no Q epilogue, 8 = no chunk loads
#endif

namespace pmt {

namespace {

constexpr int SN = 128;               // columns handled (smaller instances are zero padded)
#ifndef PMT_BS_KC
#define PMT_BS_KC 64
#endif
constexpr int SKC = PMT_BS_KC;        // rows per LDS chunk (64: 66 KB of LDS; 128 = a whole instance measured 18 % slower)
constexpr int SGP = SKC + 17;         // pitch = 17 (mod 32) doubles: the 16 columns x 2 k of half a wave land on 32 distinct bank pairs (SKC + 1 = 1 mod 32 does not)
constexpr int SPITCH = 129;           // epilogue staging pitch

typedef double f64x2 __attribute__((ext_vector_type(2)));

struct SmallArgs {
    const double *A; int64_t lda, rows, cols, strideA;
    const double *b; int64_t strideb; int sign;
    double *out_q, *out_lin, *out_const; int64_t out_stride;
    int64_t B;
    int vec_in;
    // optional constraint block of the same instance: Cm (m x cols, column-major) -> out_C row-major, out_d[i] = 0.0 (+|-) d[i]
    const double *Cm; int64_t m; const double *d; int sign_d; double *out_C, *out_d;
};

// wave w computes sub-tiles (tm = S_TM1[w][i], tn = S_TN1[w]) for i < 4 and (S_TM2[w], S_TN2[w]); a sub-tile that is not needed
// (valid bit clear) repeats a needed one and is not written
__device__ __constant__ const signed char S_TN1[8] = {7, 7, 6, 5, 4, 3, 6, 2};
__device__ __constant__ const signed char S_TM1[8][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 2, 3}, {0, 1, 2, 3}, {0, 1, 2, 3}, {0, 1, 2, 3}, {4, 5, 6, 6}, {0, 1, 2, 2}};
__device__ __constant__ const unsigned char S_VALID1[8] = {15, 15, 15, 15, 15, 15, 7, 7};
__device__ __constant__ const signed char S_TN2[8] = {5, 5, 1, 1, 4, 3, 4, 0};
__device__ __constant__ const signed char S_TM2[8] = {4, 5, 0, 1, 0, 0, 4, 0};
__device__ __constant__ const unsigned char S_VALID2[8] = {1, 1, 1, 1, 0, 0, 1, 1};

}  // namespace

// NW waves per workgroup; wave w plays the roles w, w + NW, ... of the 8-role table (roles w and w + 4 sit on the same SIMD in the
// 8-wave layout, so a 4-wave layout would keep the per-SIMD balance).  Shipped: NW = 8, one 512-thread workgroup per CU.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void batch_small_kernel(SmallArgs p) {
    constexpr int NT = NW * 64;
    constexpr int NR = 8 / NW;                               // roles per wave
    constexpr int NCLS = NT / 128;                           // row classes of the q accumulation
    __shared__ double panel[SN * SGP];                      // the chunk; reused by the epilogue for half a tile (64 x SPITCH)
    __shared__ double cvec[SKC];
    __shared__ double cvec_q[NT];
    static_assert(SN * SGP >= 64 * SPITCH, "LDS buffer must hold the epilogue staging tile");
    constexpr int KP = SKC / 2;                              // 16-byte pieces per column
    constexpr int CPP = NT / KP;                             // columns per pass
    constexpr int NP = SN / CPP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, lk = lane >> 4;
    const int kp = tid % KP, cc0 = tid / KP;
    const bool fast = p.vec_in && p.cols == SN && (p.rows % SKC) == 0 && p.rows > 0;
    const int nchunk = (int)max((int64_t)1, (p.rows + SKC - 1) / SKC);
    const int64_t n = p.cols;

    // (instance, chunk) items are software pipelined: the global loads of the NEXT item are in flight (in registers) while the
    // current chunk is multiplied — all workgroups run in step, so without this the HBM phase and the MFMA phase alternate chip-wide
    // instead of overlapping.  Out-of-range pieces read a clamped address and are replaced by zero: no branches around the loads.
    f64x2 v[NP];
    double cval = 0.0;
    auto load_chunk = [&](int64_t inst, int64_t i0) {
        const double *A = p.A + inst * p.strideA;
        const int64_t row = i0 + 2 * kp;
        if (PMT_BS_SKIP & 8) {
#pragma unroll
            for (int q = 0; q < NP; ++q) { v[q].x = 1.0; v[q].y = 2.0; }
        } else if (fast) {
#pragma unroll
            for (int q = 0; q < NP; ++q) v[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)(cc0 + CPP * q) * p.lda + row);
        } else {
            const int64_t r0 = min(row, max(p.rows - 1, (int64_t)0)), r1 = min(row + 1, max(p.rows - 1, (int64_t)0));
            const bool ok0 = row < p.rows, ok1 = row + 1 < p.rows;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int col = cc0 + CPP * q;
                const double *src = A + (int64_t)min((int64_t)col, max(p.cols - 1, (int64_t)0)) * p.lda;
                const bool okc = col < p.cols && p.rows > 0;
                const double x = okc ? src[r0] : 0.0, y = okc ? src[r1] : 0.0;
                v[q].x = (okc && ok0) ? x : 0.0;
                v[q].y = (okc && ok1) ? y : 0.0;
            }
        }
        cval = 0.0;
        if (tid < SKC) {
            const int64_t rr = i0 + tid;
            if (p.b && p.sign && rr < p.rows) cval = signed_const(p.b[inst * p.strideb + rr], p.sign);
        }
    };

    double acc[NR][5][4];
    double qpart = 0.0;                                      // thread t: column t & 127, rows (t >> 7) + NCLS*u of every chunk
    int64_t inst = blockIdx.x;
    int ch = 0;
    if (inst < p.B) load_chunk(inst, 0);
    while (inst < p.B) {
        __syncthreads();                                     // the previous chunk / epilogue is done with the buffers
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double *d = panel + (cc0 + CPP * q) * SGP + 2 * kp;
            d[0] = v[q].x; d[1] = v[q].y;
        }
        if (tid < SKC) cvec[tid] = cval;
        __syncthreads();
        int64_t ninst = inst;
        int nch = ch + 1;
        if (nch == nchunk) { nch = 0; ninst = inst + gridDim.x; }
        if (ninst < p.B) load_chunk(ninst, (int64_t)nch * SKC);
        if (ch == 0) {
#pragma unroll
            for (int ro = 0; ro < NR; ++ro)
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ro][i][r] = 0.0;
            qpart = 0.0;
        }
        // ---- q from the chunk (vector ALU; runs beside the other waves' MFMAs): thread t owns column t & 127 and the rows
        // k = (t >> 7) mod NCLS — a wave reads 64 consecutive columns at one k (pitch 65: conflict-free), c_k is an LDS broadcast
        if (!(PMT_BS_SKIP & 2)) {
            const double *colp = panel + (tid & 127) * SGP + (tid >> 7);
#pragma unroll 4
            for (int u = 0; u < SKC / NCLS; ++u) qpart += cvec[(tid >> 7) + NCLS * u] * colp[NCLS * u];
        }
        // ---- contraction: 16 k-steps of 4 rows
        if (!(PMT_BS_SKIP & 1)) {
            int rc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rc[r] = (((((lm >> 2) + r) & 3) << 2) | (lm & 3)) * SGP;     // column group rotated by r blocks
            const double *pa[NR][5], *pb1[NR], *pb2[NR];
#pragma unroll
            for (int ro = 0; ro < NR; ++ro) {
                const int role = wave + ro * NW;
#pragma unroll
                for (int i = 0; i < 4; ++i) pa[ro][i] = panel + (S_TM1[role][i] * 16 + lm) * SGP + lk;
                pa[ro][4] = panel + (S_TM2[role] * 16 + lm) * SGP + lk;
                pb1[ro] = panel + (S_TN1[role] * 16) * SGP + lk;
                pb2[ro] = panel + (S_TN2[role] * 16) * SGP + lk;
            }
            // the unused sub-tile slots of a role (group-1 slot 3 of roles 6, 7; the group-2 slot of roles 4, 5) are compiled OUT per
            // role class instead of multiplied and discarded: 9 useful sub-tiles per SIMD, not 10.  The class is wave-uniform and
            // chosen outside the k loop, whose body stays branch-free.
            auto contract = [&](auto has4_t, auto has2_t) {
                constexpr bool H4 = decltype(has4_t)::value, H2 = decltype(has2_t)::value;
#pragma unroll 2
                for (int ks = 0; ks < SKC / 4; ++ks) {
#pragma unroll
                    for (int ro = 0; ro < NR; ++ro) {
                        double a[5], b1[4], b2[4];
#pragma unroll
                        for (int i = 0; i < 3; ++i) a[i] = pa[ro][i][ks * 4];
                        if (H4) a[3] = pa[ro][3][ks * 4];
                        if (H2) a[4] = pa[ro][4][ks * 4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { b1[r] = pb1[ro][rc[r] + ks * 4]; if (H2) b2[r] = pb2[ro][rc[r] + ks * 4]; }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int i = 0; i < (H4 ? 4 : 3); ++i)
                                acc[ro][i][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b1[r], acc[ro][i][r], 0, 0, 0);
                            if (H2) acc[ro][4][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[4], b2[r], acc[ro][4][r], 0, 0, 0);
                        }
                    }
                }
            };
            using T = std::true_type;
            using F = std::false_type;
            if (NR == 1 && wave >= 6) contract(F{}, T{});
            else if (NR == 1 && wave >= 4) contract(T{}, F{});
            else contract(T{}, T{});
        }
        // ---- outputs after the last chunk of an instance
        if (ch == nchunk - 1) {
            double *outq = p.out_q + inst * p.out_stride;
            if (p.Cm) {                                      // constraint block: a 16 KB transposition, read through L2
                const double *Ci = p.Cm + inst * p.m * n;
                double *oc = p.out_C + inst * p.out_stride;
                for (int64_t e = tid; e < p.m * n; e += NT) {
                    const int64_t row = e / n, col = e - row * n;
                    oc[e] = Ci[col * p.m + row];
                }
                for (int64_t i = tid; i < p.m; i += NT) p.out_d[inst * p.out_stride + i] = signed_const(p.d[inst * p.m + i], p.sign_d);
            }
            __syncthreads();                                 // q: add the row classes of a column (through LDS), 2x
            cvec_q[tid] = qpart;
            __syncthreads();
            if (tid < 128 && tid < p.cols) {
                double sum = cvec_q[tid];
#pragma unroll
                for (int c = 1; c < NCLS; ++c) sum = sum + cvec_q[tid + 128 * c];
                p.out_lin[inst * p.out_stride + tid] = 2 * sum;
            }
            const int i_ = lane >> 4, bq = (lane >> 2) & 3, j_ = lane & 3;       // accumulator element -> (row, col) inside a sub-tile
            for (int h = 0; h < ((PMT_BS_SKIP & 4) ? 0 : 2); ++h) {
                __syncthreads();
#pragma unroll
                for (int ro = 0; ro < NR; ++ro) {
                    const int role = wave + ro * NW;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const int tm = i < 4 ? S_TM1[role][i] : S_TM2[role];
                        const int tn = i < 4 ? S_TN1[role] : S_TN2[role];
                        const bool valid = i < 4 ? ((S_VALID1[role] >> i) & 1) : (S_VALID2[role] & 1);
                        if (valid && (tm >> 2) == h) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = tm * 16 + 4 * bq + i_, col = tn * 16 + 4 * ((bq + r) & 3) + j_;
                                panel[(row - 64 * h) * SPITCH + col] = 2 * acc[ro][i][r];
                            }
                        }
                    }
                }
                __syncthreads();
                for (int row = wave; row < 64; row += NW) {
                    const int64_t j = 64 * h + row;
                    if (j >= n) break;
                    const int64_t term0 = j * n - (j * (j - 1)) / 2;
                    for (int64_t k = j + lane; k < n; k += 64) outq[term0 + (k - j)] = panel[row * SPITCH + k];
                }
            }
        }
        inst = ninst;
        ch = nch;
    }
}

bool batch_small_enabled() {
    static const bool on = [] { const char *e = getenv("PMT_BATCH_SMALL"); return !(e && e[0] == '0'); }();
    return on;
}

int launch_batch_small(const double *A, int64_t lda, int64_t rows, int64_t cols, int64_t strideA, const double *b, int64_t strideb, int sign,
                       int64_t B, double *out_q, double *out_lin, double *out_const, int64_t out_stride,
                       const double *Cm, int64_t m, const double *d, int sign_d, double *out_C, double *out_d, hipStream_t s) {
    SmallArgs p;
    p.Cm = (m > 0 && cols > 0) ? Cm : nullptr; p.m = m; p.d = d; p.sign_d = sign_d; p.out_C = out_C; p.out_d = out_d;
    p.A = A; p.lda = lda; p.rows = rows; p.cols = cols; p.strideA = strideA; p.b = b; p.strideb = strideb; p.sign = sign;
    p.out_q = out_q; p.out_lin = out_lin; p.out_const = out_const; p.out_stride = out_stride; p.B = B;
    p.vec_in = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0 && (strideA & 1) == 0) ? 1 : 0;
    // (a 4-wave instantiation — two workgroups per CU — measured 1.16 ms against 0.72 ms for this one; profiles/r01d_side_stream.txt)
    PMT_LAUNCH_NAMED("batch_small_kernel", batch_small_kernel<8>, dim3((unsigned)std::min<int64_t>(B, 4096)), dim3(512), 0, s, p);
    return check_launch("batch_small_kernel");
}

}  // namespace pmt
