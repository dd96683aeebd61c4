// Canonical least-squares objective: canonicalize!(residual . residual) for residual = A*x (+|-) b, fused with
// the MOI copy.  out_quad = upper triangle (row-major) of 2*A'A as QuadraticTerms, out_lin = 2*A'c, out_const = c'c
// (SURVEY.md Appendix A.3).  The only arithmetic-heavy node of the path: r*n*(n+1) fp64 FLOP -> f64 MFMA.
//
// Reference semantics replaced: _vecdot!/muladd! literal expansion (src/functions.jl:702-709,548-576) followed by
// canonicalize! (src/functions.jl:381-386, sort_and_combine! src/util.jl:9-26) and
// update!(::MOI.ScalarQuadraticFunction, ...) (src/moi_interop.jl:45-62).  The reference sums duplicates in
// QuickSort order; here the contraction index runs in row order inside v_mfma_f64_16x16x4_f64 — coefficients
// agree to rounding (tests: <= 1e-12 relative), indices exactly.
//
// Kernel shape (gfx950): 128x128 output tile per 256-thread workgroup (2x2 waves, each 64x64 = 4x4 MFMA tiles,
// 64 fp64 accumulators per lane).  A is column-major, so both MFMA operands are K-contiguous column panels
// [128 columns][BK rows]; panels are staged global -> registers -> LDS (double buffered, one barrier per stage)
// with 16-byte loads along K.  f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4*reg
// (cdna_hip_programming.md §3 — differs from the f32/bf16 map).  Only tiles with jb <= kb are computed.
#include <cstring>

#include "common.h"

namespace pmt {

int launch_seq_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *out, hipStream_t s);
size_t gram_sk_workspace_bytes(int64_t rows, int64_t cols);
int launch_gram_sk(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *varmap, int moi,
                   pmt_quadratic_term *out_quad, double *out_csc, double alpha, void *workspace, hipStream_t s);

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int GT = 128;          // output tile edge
constexpr int BK = 16;           // contraction depth per LDS stage
constexpr int GP = BK + 2;       // LDS pitch (doubles) of one column of a panel; even keeps 16-byte alignment

struct GramArgs {
    const double *A; int64_t lda, rows, cols;
    const int64_t *xvar; const int64_t *varmap; int moi;
    QT *out_quad;
    int ntiles;      // tiles per side
    int vec_in;
};

// upper-triangular tile index -> (jb, kb), jb <= kb, row-major over the triangle
__device__ __forceinline__ void tri_unrank(int t, int nt, int &jb, int &kb) {
    // row jb starts at S(jb) = jb*nt - jb*(jb-1)/2
    int j = (int)((2.0 * nt + 1.0 - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)t)) * 0.5);
    if (j < 0) j = 0;
    if (j > nt - 1) j = nt - 1;
    while (j > 0 && (j * nt - j * (j - 1) / 2) > t) --j;
    while (j + 1 < nt && ((j + 1) * nt - (j + 1) * j / 2) <= t) ++j;
    jb = j;
    kb = j + (t - (j * nt - j * (j - 1) / 2));
}

__device__ __forceinline__ void load_panel(const GramArgs &g, int64_t c0, int64_t i0, f64x2 (&reg)[4], int tid) {
    // 128 columns x BK rows; thread -> (column = tid/8 + 32*p, row pair = (tid%8)*2)
    const int rr = (tid & 7) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t col = c0 + (tid >> 3) + 32 * p;
        const int64_t row = i0 + rr;
        f64x2 v; v.x = 0.0; v.y = 0.0;
        if (col < g.cols) {
            const double *src = g.A + col * g.lda + row;
            if (g.vec_in && row + 1 < g.rows) v = *reinterpret_cast<const f64x2 *>(src);
            else {
                if (row < g.rows) v.x = src[0];
                if (row + 1 < g.rows) v.y = src[1];
            }
        }
        reg[p] = v;
    }
}
__device__ __forceinline__ void store_panel(double *panel, const f64x2 (&reg)[4], int tid) {
    const int rr = (tid & 7) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int col = (tid >> 3) + 32 * p;
        *reinterpret_cast<f64x2 *>(panel + col * GP + rr) = reg[p];
    }
}

// M4 = true : v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction).  Measured on MI355X
//             (tools/mfma_f64_peak.hip): 17 cycles/instruction = 72.5 TFLOP/s, i.e. the datasheet FP64-matrix
//             rate, whereas v_mfma_f64_16x16x4_f64 issues every 138 cycles = 34.9 TFLOP/s.  The hardware
//             ignores cbsz/abid on f64 MFMA (tools/mfma_probe2.hip), so the 16 (row group, column group) pairs
//             of a 16x16 tile are covered by 4 instructions whose B operand is read from LDS with the column
//             groups rotated by s = 0..3 blocks.  Operand lane maps (tools/mfma_probe.hip):
//               A: lane = i + 4b + 16k   B: lane = j + 4b + 16k   D: lane = j + 4b + 16i   (block b, 4x4 tile)
//             so A/B registers are those of the 16x16x4 form (row|col = lane & 15, k = lane >> 4) and
//             acc[tm][tn][s] of lane l holds C[16tm + 4b + i][16tn + 4((b+s)&3) + j], i = l>>4, b = (l>>2)&3, j = l&3.
// M4 = false: v_mfma_f64_16x16x4_f64, acc[tm][tn][v] = C[16tm + (l>>4) + 4v][16tn + (l&15)].
template <bool M4>
__global__ __launch_bounds__(256, M4 ? 1 : 2) void quad_gram_kernel(GramArgs g) {
    __shared__ double lds[2][2][GT * GP];   // [buffer][panel J/K][col*GP + k]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-contiguous chunks (blocks round-robin over 8 XCDs)
    int jb, kb;
    tri_unrank(bid, g.ntiles, jb, kb);
    const int64_t j0 = (int64_t)jb * GT, k0 = (int64_t)kb * GT;
    const bool diag = (jb == kb);

    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};

    f64x2 rj[4], rk[4];
    const int nstage = (int)((g.rows + BK - 1) / BK);
    load_panel(g, j0, 0, rj, tid);
    if (!diag) load_panel(g, k0, 0, rk, tid);
    store_panel(lds[0][0], rj, tid);
    if (!diag) store_panel(lds[0][1], rk, tid);
    __syncthreads();

    const int lm = lane & 15, lk = lane >> 4;
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        if (s + 1 < nstage) {
            load_panel(g, j0, (int64_t)(s + 1) * BK, rj, tid);
            if (!diag) load_panel(g, k0, (int64_t)(s + 1) * BK, rk, tid);
        }
        const double *pj = lds[cur][0] + (wr * 64 + lm) * GP + lk;
        const double *pkbase = lds[cur][diag ? 0 : 1] + (wc * 64) * GP + lk;
        const double *pk = pkbase + lm * GP;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = pj[t * 16 * GP + ks * 4];
            if (M4) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    double b[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int rc = ((((lm >> 2) + s) & 3) << 2) | (lm & 3);        // column group rotated by s blocks
                        b[s] = pkbase[(tn * 16 + rc) * GP + ks * 4];
                    }
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc[tm][tn][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tm], b[s], acc[tm][tn][s], 0, 0, 0);
                }
            } else {
                double b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = pk[t * 16 * GP + ks * 4];
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        if (s + 1 < nstage) {
            store_panel(lds[cur ^ 1][0], rj, tid);
            if (!diag) store_panel(lds[cur ^ 1][1], rk, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: 2*acc -> QuadraticTerm at the canonical upper-triangular position
    const int64_t n = g.cols;
    u64 *out = reinterpret_cast<u64 *>(g.out_quad);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int64_t j, k;
                if (M4) {
                    const int b = (lane >> 2) & 3;
                    j = j0 + wr * 64 + tm * 16 + 4 * b + lk;
                    k = k0 + wc * 64 + tn * 16 + 4 * ((b + v) & 3) + (lane & 3);
                } else {
                    j = j0 + wr * 64 + tm * 16 + lk + 4 * v;
                    k = k0 + wc * 64 + tn * 16 + lm;
                }
                if (k >= n || j >= n || j > k) continue;
                const int64_t kv = g.xvar[k];
                const u64 kvm = (u64)(g.moi ? map_var(g.varmap, kv) : kv);
                const int64_t jv = g.xvar[j];
                double c = acc[tm][tn][v];
                if (g.moi || j != k) c = 2 * c;          // off-diagonal: (j,k)+(k,j) combined; diagonal: MOI doubling
                const int64_t pos = j * n - (j * (j - 1)) / 2 + (k - j);
                u64 *p = out + pos * 3;
                p[0] = (u64)__double_as_longlong(c);
                p[1] = (u64)(g.moi ? map_var(g.varmap, jv) : jv);
                p[2] = kvm;
            }
        }
    }
}

// out_lin[j] = (2 * sum_i c_i * A[i,j], vm[xvar[j]]),  c_i = 0.0 (+|-) b[i]; one wave per column (coalesced along i)
__global__ __launch_bounds__(256) void gram_linear_kernel(const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                                          const int64_t *__restrict__ xvar, const double *__restrict__ b, int sign,
                                                          int moi, const int64_t *__restrict__ varmap, LT *__restrict__ out_lin) {
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const int lane = threadIdx.x & 63;
    const double *a = A + col * lda;
    double acc = 0.0;
    if (b && sign)
        for (int64_t i = lane; i < rows; i += 64) acc += signed_const(b[i], sign) * a[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) {
        LT t;
        t.coeff = 2 * acc;
        const int64_t v = xvar[col];
        t.var = moi ? map_var(varmap, v) : v;
        out_lin[col] = t;
    }
}

}  // namespace pmt

namespace pmt {

// one non-blocking side stream + fork/join events per device, created on first use (PMT_GRAM_SIDE_STREAM=0 disables)
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static SideStream *side_stream() {
    static const bool enabled = [] { const char *e = getenv("PMT_GRAM_SIDE_STREAM"); return !(e && e[0] == '0'); }();
    if (!enabled) return nullptr;
    static SideStream per_device[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
    SideStream &ss = per_device[dev];
    if (!ss.stream) {
        if (hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ss.stream = nullptr;
            return nullptr;
        }
    }
    return &ss;
}

}  // namespace pmt

using namespace pmt;

extern "C" size_t pmt_quad_gram_workspace_bytes(int64_t rows, int64_t cols) {
    return gram_sk_workspace_bytes(rows, cols);
}

// the whole node: contraction on the main stream, q = 2A'c and c'c on a side stream.  out_quad (term structs) and out_csc (solver
// values, alpha-scaled) are independent optional outputs of the same contraction.
static int gram_node(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                     int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                     double *out_const, void *workspace, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "quad_gram: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "quad_gram: lda < rows");
    PMT_REQUIRE(sign >= -1 && sign <= 1, PMT_INVALID_ARGUMENT, "quad_gram: sign must be -1, 0 or +1");
    PMT_REQUIRE(out_const, PMT_INVALID_ARGUMENT, "quad_gram: null out_const");
    PMT_REQUIRE(sign == 0 || b || rows == 0, PMT_INVALID_ARGUMENT, "quad_gram: sign != 0 needs b");
    if (cols > 0) PMT_REQUIRE(xvar && (out_quad || out_csc) && out_lin && (A || rows == 0), PMT_INVALID_ARGUMENT, "quad_gram: null pointer");
    PMT_REQUIRE(cols < (int64_t)GT * 46000, PMT_DIMENSION_MISMATCH, "quad_gram: too many columns");
    return dispatch(stream, [=](hipStream_t s) {
        // fork: the two small reductions of this node (q = 2 A'c, HBM-bound; c'c, a serial chain) run on a side stream while
        // the MFMA-bound contraction owns the main stream; join before returning control of `s`.  Legal under stream capture.
        SideStream *side = side_stream();
        hipStream_t s2 = s;
        if (side) {
            PMT_HIP_CHECK(hipEventRecord(side->fork, s));
            PMT_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
            s2 = side->stream;
        }
        int rc = PMT_OK;
        if (cols > 0) {
            PMT_LAUNCH(gram_linear_kernel, dim3((unsigned)cdiv(cols, 4)), dim3(256), 0, s2, A, lda, rows, cols, xvar, b, sign, moi, varmap, out_lin);
            rc = check_launch("gram_linear_kernel");
        }
        if (!rc) {
            if (b && sign && rows > 0) rc = launch_seq_dot(b, sign, b, sign, rows, out_const, s2);
            else if (hipMemsetAsync(out_const, 0, sizeof(double), s2) != hipSuccess) rc = fail(PMT_HIP_ERROR, "hipMemsetAsync(out_const)");
        }
        if (side) PMT_HIP_CHECK(hipEventRecord(side->join, side->stream));
        if (!rc && cols > 0) {
            GramArgs g;
            g.A = A; g.lda = lda; g.rows = rows; g.cols = cols; g.xvar = xvar; g.varmap = varmap; g.moi = moi; g.out_quad = out_quad;
            g.ntiles = (int)cdiv(cols, GT);
            g.vec_in = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0) ? 1 : 0;
            const int nblk = g.ntiles * (g.ntiles + 1) / 2;
            // PMT_GRAM_IMPL: "sk" (default) stream-K persistent kernel on the 4x4x4_4b MFMA (gram_sk.hip);
            // "tiles16" / "tiles4": one workgroup per tile on the 16x16x4 / 4x4x4_4b MFMA (first-round kernels, kept for A/B:
            // 1.85 ms / 2.56 ms at n = r = 4096, profiles/r01a_*).
            static const int impl = [] {
                const char *e = getenv("PMT_GRAM_IMPL");
                if (e && !strcmp(e, "tiles16")) return 1;
                if (e && !strcmp(e, "tiles4")) return 2;
                return 0;
            }();
            if (impl == 0) {
                rc = launch_gram_sk(A, lda, rows, cols, xvar, varmap, moi, out_quad, out_csc, alpha, workspace, s);
            } else if (out_csc) {
                rc = fail(PMT_INVALID_ARGUMENT, "quad_gram_csc: only the stream-K implementation writes CSC values (unset PMT_GRAM_IMPL)");
            } else {
                if (impl == 1) PMT_LAUNCH_NAMED("quad_gram_kernel", quad_gram_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, g);
                else PMT_LAUNCH_NAMED("quad_gram_kernel", quad_gram_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, g);
                rc = check_launch("quad_gram_kernel");
            }
        }
        if (side) PMT_HIP_CHECK(hipStreamWaitEvent(s, side->join, 0));
        return rc;
    });
}

extern "C" int pmt_quad_gram_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                 int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                                 void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_quad, PMT_INVALID_ARGUMENT, "quad_gram: null out_quad");
    return gram_node(A, lda, rows, cols, xvar, b, sign, moi, varmap, out_quad, nullptr, 1.0, out_lin, out_const, workspace, stream);
}

extern "C" int pmt_quad_gram_csc_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                                     const int64_t *varmap, double alpha, double *out_P_values, pmt_quadratic_term *out_quad,
                                     pmt_linear_term *out_lin, double *out_const, void *workspace, void *stream) {
    if (cols > 0) PMT_REQUIRE(out_P_values, PMT_INVALID_ARGUMENT, "quad_gram_csc: null out_P_values");
    return gram_node(A, lda, rows, cols, xvar, b, sign, 1, varmap, out_quad, out_P_values, alpha, out_lin, out_const, workspace, stream);
}
