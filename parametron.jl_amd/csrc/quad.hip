// Quadratic nodes, literal (uncombined) forms: every output term is one product written once.
// Write-bound streaming kernels: 24-byte QuadraticTerm AoS output assembled as 16-byte chunks so that the
// bulk of the stores are global_store_dwordx4 regardless of the 24-byte element size.
//
// Reference loops replaced (see include/parametron_hip.h):
//   _vecdot!(::QuadraticFunction, x, y)  src/functions.jl:702-709  over  muladd! :548-576 / :537-546
//   _vecdot! Variable/LinearTerm form    src/functions.jl:689-700
//   bilinearmul!                         src/functions.jl:840-858
//   update!(::MOI.ScalarQuadraticFunction, f, varmap)  src/moi_interop.jl:45-62 (fused when moi != 0)
#include "common.h"

namespace pmt {

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64 d2u(double x) { return (u64)__double_as_longlong(x); }

// Cooperative write of `nterms` consecutive QuadraticTerms starting at term index `term0` of `out`
// (viewed as an array of 8-byte words).  F(term) -> (coeff bits, row, col) is evaluated per word.
// All lanes of the block must call this together.
template <typename F>
__device__ __forceinline__ void write_qt_segment(u64 *__restrict__ out, int64_t term0, int nterms, F f) {
    const int64_t base = term0 * 3;                 // first 8-byte word of the segment
    const int nwords = nterms * 3;
    const int lead = (int)((reinterpret_cast<uintptr_t>(out + base) >> 3) & 1);   // 1: segment starts in the upper half of a 16-byte slot
    if (lead && threadIdx.x == 0) out[base] = f(0, 0);
    for (int c = threadIdx.x; lead + 2 * c < nwords; c += blockDim.x) {
        const int q0 = lead + 2 * c;
        const int t0 = q0 / 3, f0 = q0 - 3 * t0;
        if (q0 + 1 < nwords) {
            const int q1 = q0 + 1;
            const int t1 = q1 / 3, f1 = q1 - 3 * t1;
            u64x2 v;
            v.x = f(t0, f0);
            v.y = f(t1, f1);
            *reinterpret_cast<u64x2 *>(out + base + q0) = v;
        } else {
            out[base + q0] = f(t0, f0);
        }
    }
}

// ---- literal expansion of x . y for Vector{AffineFunction} with uniform row lengths
constexpr int QE_BT = 1024;   // y-terms staged per block
constexpr int QE_AB = 8;      // x-terms per block

__global__ __launch_bounds__(256) void quad_expand_kernel(
    int64_t rows, const LT *__restrict__ x_terms, int64_t nx, const LT *__restrict__ y_terms, int64_t ny,
    int moi, const int64_t *__restrict__ varmap, u64 *__restrict__ out_quad) {
    __shared__ double yc[QE_BT];
    __shared__ u64 yvm[QE_BT];
    __shared__ int64_t yv[QE_BT];
    const int64_t b0 = (int64_t)blockIdx.x * QE_BT;
    const int bt = (int)min((int64_t)QE_BT, ny - b0);
    const int64_t a0 = (int64_t)blockIdx.y * QE_AB;
    for (int64_t i = blockIdx.z; i < rows; i += gridDim.z) {
        __syncthreads();
        for (int k = threadIdx.x; k < bt; k += blockDim.x) {
            const LT t = y_terms[i * ny + b0 + k];
            yc[k] = t.coeff;
            yv[k] = t.var;
            yvm[k] = (u64)(moi ? map_var(varmap, t.var) : t.var);
        }
        __syncthreads();
        const int64_t a1 = min(a0 + QE_AB, nx);
        for (int64_t a = a0; a < a1; ++a) {
            const LT xa = x_terms[i * nx + a];
            const u64 xvm = (u64)(moi ? map_var(varmap, xa.var) : xa.var);
            write_qt_segment(out_quad, (i * nx + a) * ny + b0, bt, [&](int term, int field) -> u64 {
                if (field == 0) {
                    double c = xa.coeff * yc[term];                       // functions.jl:149
                    if (moi && xa.var == yv[term]) c = 2 * c;             // moi_interop.jl:58
                    return d2u(c);
                }
                return field == 1 ? xvm : yvm[term];
            });
        }
    }
}

// affine part of muladd!(dest, x::AffineFunction, y::AffineFunction): src/functions.jl:560-573
__global__ void quad_expand_linear_kernel(int64_t rows, const LT *__restrict__ x_terms, int64_t nx, const double *__restrict__ x_consts,
                                          const LT *__restrict__ y_terms, int64_t ny, const double *__restrict__ y_consts,
                                          int moi, const int64_t *__restrict__ varmap, LT *__restrict__ out_lin) {
    const int64_t w = nx + ny;
    const int64_t total = rows * w;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t i = idx / w;
        const int64_t k = idx - i * w;
        LT t;
        double c;
        if (k < nx) { t = x_terms[i * nx + k]; c = y_consts[i]; }       // xlinear[i] * yconst  (:567)
        else { t = y_terms[i * ny + (k - nx)]; c = x_consts[i]; }       // ylinear[i] * xconst  (:571)
        LT o;
        o.coeff = c * t.coeff;                                           // term * c -> c * coeff (:159-160)
        o.var = moi ? map_var(varmap, t.var) : t.var;
        out_lin[idx] = o;
    }
}

// out = ((0 + a0*b0) + a1*b1) + ...  strictly left to right (src/functions.jl:574 under the loop of :705-707).
// sign_a / sign_b: 2 = use the array as is; -1/0/+1 = use 0.0 (+|-) value (constants of a fused A*x (+|-) b node).
// One wave, at most 16 VGPRs: the persistent Gram kernel holds 2 x 248 of a SIMD's 512 VGPRs, so only a kernel this small is
// CO-RESIDENT with it on the side stream (anything bigger waits for its workgroups to drain and lands on the critical path —
// profiles/r01d_side_stream.txt).  Lane l holds the product of element base + l; the chain runs over v_readlane broadcasts, the
// next 64 products are loaded while the current 64 are added.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <bool SAME>
__global__ __launch_bounds__(64) void seq_dot_kernel(const double *__restrict__ a, int sign_a, const double *__restrict__ b, int sign_b,
                                                     int n, double *__restrict__ out) {
    const int lane = threadIdx.x;
    auto product = [&](int i) -> double {                  // 32-bit indices: uniform base pointer + lane offset, few registers
        if (i >= n) return 0.0;
        const double av = sign_a == 2 ? a[i] : signed_const(a[i], sign_a);
        if (SAME) return av * av;
        if (sign_b == 3) return av;                        // plain sum of a (second level of launch_blocked_dot)
        const double bv = sign_b == 2 ? b[i] : signed_const(b[i], sign_b);
        return av * bv;
    };
    double acc = 0.0;
    double p = product(lane);
    for (int base = 0; base < n; base += 64) {
        const double cur = p;
        p = product(base + 64 + lane);                      // in flight during the chain below
        const int left = n - base;
        if (left >= 64) {
#pragma unroll
            for (int j = 0; j < 64; ++j) acc = acc + readlane_f64(cur, j);
        } else {
            for (int j = 0; j < left; ++j) acc = acc + readlane_f64(cur, j);
        }
    }
    if (lane == 0) *out = acc;
}

int launch_seq_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *out, hipStream_t s) {
    if (n >= ((int64_t)1 << 31) - 64) return fail(PMT_DIMENSION_MISMATCH, "seq_dot: vector too long");
    if (a == b && sign_a == sign_b) PMT_LAUNCH_NAMED("seq_dot_kernel", seq_dot_kernel<true>, dim3(1), dim3(64), 0, s, a, sign_a, b, sign_b, (int)n, out);
    else PMT_LAUNCH_NAMED("seq_dot_kernel", seq_dot_kernel<false>, dim3(1), dim3(64), 0, s, a, sign_a, b, sign_b, (int)n, out);
    return check_launch("seq_dot_kernel");
}

// Long vectors (least squares with many rows): r dependent additions are r x ~25 cycles on one wave — 15 ms at r = 10^6, far more than the
// contraction they sit beside.  Above DOT_EXACT_MAX elements the sum is taken in DOT_CHAINS interleaved chains (thread t adds the products
// of elements t, t + DOT_CHAINS, ... in order) and the chain totals are then added left to right by the one-wave kernel above.  Fixed
// order, so deterministic, but NOT the reference's sequential order: for these lengths the constant is within (n / DOT_CHAINS + DOT_CHAINS)
// * eps / 2 relative of the exact sum of the same products (positive terms), well inside the 1e-12 of the parity bar.
constexpr int DOT_EXACT_MAX = 8192;
constexpr int DOT_CHAINS = 2048;

template <bool SAME>
__global__ __launch_bounds__(256) void dot_chains_kernel(const double *__restrict__ a, int sign_a, const double *__restrict__ b, int sign_b,
                                                         int n, double *__restrict__ chains) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (int i = t; i < n; i += DOT_CHAINS) {
        const double av = sign_a == 2 ? a[i] : signed_const(a[i], sign_a);
        double p;
        if (SAME) p = av * av;
        else p = av * (sign_b == 2 ? b[i] : signed_const(b[i], sign_b));
        acc = acc + p;
    }
    chains[t] = acc;
}

size_t blocked_dot_scratch_doubles() { return DOT_CHAINS; }

// launch_seq_dot for vectors of any length; `scratch` (DOT_CHAINS doubles, device) is only touched above DOT_EXACT_MAX elements
// chained != 0: take the chained form whatever the length (the caller's cost model, gram.hip: a sequential chain that would outlast the
// contraction it sits beside)
int launch_blocked_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *scratch, double *out, hipStream_t s, int chained) {
    if (!scratch || (n <= DOT_EXACT_MAX && !chained)) return launch_seq_dot(a, sign_a, b, sign_b, n, out, s);
    if (n >= ((int64_t)1 << 31) - DOT_CHAINS) return fail(PMT_DIMENSION_MISMATCH, "blocked_dot: vector too long");
    if (a == b && sign_a == sign_b) PMT_LAUNCH_NAMED("dot_chains_kernel", dot_chains_kernel<true>, dim3(DOT_CHAINS / 256), dim3(256), 0, s, a, sign_a, b, sign_b, (int)n, scratch);
    else PMT_LAUNCH_NAMED("dot_chains_kernel", dot_chains_kernel<false>, dim3(DOT_CHAINS / 256), dim3(256), 0, s, a, sign_a, b, sign_b, (int)n, scratch);
    if (int rc = check_launch("dot_chains_kernel")) return rc;
    return launch_seq_dot(scratch, 2, scratch, 3, DOT_CHAINS, out, s);
}

// ---- bilinearmul!: one block row per x index.  The coefficients of the NEXT row are fetched into registers while the current row's
// terms are written, and the LDS row buffer is double buffered: one barrier per row and no exposed load latency.
__global__ __launch_bounds__(256) void bilinear_kernel(const double *__restrict__ Q, int64_t ldq, int64_t nxr, int64_t ny,
                                                       const int64_t *__restrict__ xvar, const int64_t *__restrict__ yvar,
                                                       int moi, const int64_t *__restrict__ varmap, u64 *__restrict__ out_quad) {
    constexpr int NPT = QE_BT / 256;
    __shared__ double qc[2][QE_BT];
    __shared__ u64 yvm[QE_BT];
    __shared__ int64_t yv[QE_BT];
    const int64_t b0 = (int64_t)blockIdx.x * QE_BT;
    const int bt = (int)min((int64_t)QE_BT, ny - b0);
    for (int k = threadIdx.x; k < bt; k += blockDim.x) {
        const int64_t v = yvar[b0 + k];
        yv[k] = v;
        yvm[k] = (u64)(moi ? map_var(varmap, v) : v);
    }
    // Q[k]: column-major LINEAR index lin = r*ny + b0 + k of the nxr x ny matrix (:853).  One 64-bit division per block row
    // (uniform) instead of a div/mod per element.
    auto fetch_row = [&](int64_t r, double (&v)[NPT]) {
        const int64_t lin0 = r * ny + b0;
        const int64_t col0 = lin0 / nxr, row0 = lin0 - col0 * nxr;
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int k = threadIdx.x + 256 * u;
            v[u] = 0.0;
            if (k < bt) {
                int64_t qc_row = row0 + k, qc_col = col0;
                if (nxr >= QE_BT) {
                    if (qc_row >= nxr) { qc_row -= nxr; ++qc_col; }  // k < QE_BT <= nxr: at most one wrap
                } else {
                    qc_col += qc_row / nxr;
                    qc_row = qc_row % nxr;
                }
                v[u] = Q[qc_col * ldq + qc_row];
            }
        }
    };
    double v[NPT];
    int64_t r = blockIdx.y;
    if (r < nxr) fetch_row(r, v);
    int cur = 0;
    for (; r < nxr; r += gridDim.y) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) qc[cur][threadIdx.x + 256 * u] = v[u];
        __syncthreads();                                         // row r is in qc[cur]; the other buffer's readers (row r - stride) are done
        if (r + gridDim.y < nxr) fetch_row(r + gridDim.y, v);     // in flight while this row is written
        const int64_t xv = xvar[r];
        const u64 xvm = (u64)(moi ? map_var(varmap, xv) : xv);
        const double *row = qc[cur];
        write_qt_segment(out_quad, r * ny + b0, bt, [&](int term, int field) -> u64 {
            if (field == 0) {
                double c = row[term];
                if (moi && xv == yv[term]) c = 2 * c;
                return d2u(c);
            }
            return field == 1 ? xvm : yvm[term];
        });
        cur ^= 1;
    }
}

__global__ void vecdot_terms_kernel(int64_t n, const double *__restrict__ xc, const int64_t *__restrict__ xvar,
                                    const double *__restrict__ yc, const int64_t *__restrict__ yvar, int moi,
                                    const int64_t *__restrict__ varmap, QT *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = xc ? xc[i] : 1.0, b = yc ? yc[i] : 1.0;
    double c = a * b;                                                   // functions.jl:146-149
    const int64_t xv = xvar[i], yv = yvar[i];
    if (moi && xv == yv) c = 2 * c;
    QT o;
    o.coeff = c;
    o.row = moi ? map_var(varmap, xv) : xv;
    o.col = moi ? map_var(varmap, yv) : yv;
    out[i] = o;
}

__global__ void vecdot_affs_vars_kernel(int64_t rows, const LT *__restrict__ x_terms, int64_t L, const double *__restrict__ x_consts,
                                        const int64_t *__restrict__ yvar, int moi, const int64_t *__restrict__ varmap,
                                        QT *__restrict__ out_quad, LT *__restrict__ out_lin) {
    const int64_t total = rows * L;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total + rows; idx += stride) {
        if (idx < total) {
            const int64_t i = idx / L;
            const LT t = x_terms[idx];
            const int64_t yv = yvar[i];
            QT o;
            o.coeff = (moi && t.var == yv) ? 2 * t.coeff : t.coeff;      // LinearTerm * Variable (:147)
            o.row = moi ? map_var(varmap, t.var) : t.var;
            o.col = moi ? map_var(varmap, yv) : yv;
            out_quad[idx] = o;
        } else {
            const int64_t i = idx - total;
            LT o;
            o.coeff = x_consts[i];                                        // x.constant[] * y (:543)
            o.var = moi ? map_var(varmap, yvar[i]) : yvar[i];
            out_lin[i] = o;
        }
    }
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_quad_expand_f64(int64_t rows, const pmt_linear_term *x_terms, int64_t nx, const double *x_consts,
                                   const pmt_linear_term *y_terms, int64_t ny, const double *y_consts, int moi, const int64_t *varmap,
                                   pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const, void *stream) {
    PMT_REQUIRE(rows >= 0 && nx >= 0 && ny >= 0, PMT_DIMENSION_MISMATCH, "quad_expand: negative dimension");
    PMT_REQUIRE(out_const, PMT_INVALID_ARGUMENT, "quad_expand: null out_const");
    if (rows > 0) {
        PMT_REQUIRE(x_consts && y_consts, PMT_INVALID_ARGUMENT, "quad_expand: null constants");
        PMT_REQUIRE((nx == 0 || x_terms) && (ny == 0 || y_terms), PMT_INVALID_ARGUMENT, "quad_expand: null terms");
        PMT_REQUIRE(nx * ny == 0 || out_quad, PMT_INVALID_ARGUMENT, "quad_expand: null out_quad");
        PMT_REQUIRE(nx + ny == 0 || out_lin, PMT_INVALID_ARGUMENT, "quad_expand: null out_lin");
    }
    SmallNode nd;
    nd.op = SOP_QUAD_EXPAND; nd.moi = moi; nd.d[0] = rows; nd.d[1] = nx; nd.d[2] = ny;
    nd.in[0] = x_terms; nd.in[1] = x_consts; nd.in[2] = y_terms; nd.in[3] = y_consts; nd.in[4] = varmap;
    nd.out[0] = out_quad; nd.out[1] = out_lin; nd.out[2] = out_const;
    // the node's constant is a sequential chain on one thread of the interpreter: short vectors only
    nd.work = rows > 2048 ? SMALL_NODE_WORK_MAX + 1 : rows * nx * ny + rows * (nx + ny) + 16 * rows;
    return dispatch(stream, [=](hipStream_t s) {
        if (rows > 0 && nx > 0 && ny > 0) {
            dim3 grid((unsigned)cdiv(ny, QE_BT), (unsigned)cdiv(nx, QE_AB), (unsigned)std::min<int64_t>(rows, 65535));
            PMT_LAUNCH(quad_expand_kernel, grid, dim3(256), 0, s, rows, x_terms, nx, y_terms, ny, moi, varmap,
                               reinterpret_cast<u64 *>(out_quad));
            int rc = check_launch("quad_expand_kernel");
            if (rc) return rc;
        }
        if (rows > 0 && nx + ny > 0) {
            const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(rows * (nx + ny), 256), 256 * 8);
            PMT_LAUNCH(quad_expand_linear_kernel, dim3(blocks), dim3(256), 0, s, rows, x_terms, nx, x_consts, y_terms, ny, y_consts,
                               moi, varmap, out_lin);
            int rc = check_launch("quad_expand_linear_kernel");
            if (rc) return rc;
        }
        return launch_seq_dot(x_consts, 2, y_consts, 2, rows, out_const, s);
    }, nd);
}

extern "C" int pmt_bilinear_f64(const double *Q, int64_t ldq, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *yvar, int moi,
                                const int64_t *varmap, pmt_quadratic_term *out_quad, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "bilinear: negative dimension");
    PMT_REQUIRE(ldq >= rows, PMT_DIMENSION_MISMATCH, "bilinear: ldq < rows");
    if (rows == 0 || cols == 0) return PMT_OK;
    PMT_REQUIRE(Q && xvar && yvar && out_quad, PMT_INVALID_ARGUMENT, "bilinear: null pointer");
    SmallNode nd;
    nd.op = SOP_BILINEAR; nd.moi = moi; nd.d[0] = ldq; nd.d[1] = rows; nd.d[2] = cols; nd.in[0] = Q; nd.in[1] = xvar; nd.in[2] = yvar; nd.in[3] = varmap;
    nd.out[0] = out_quad; nd.work = rows * cols;
    return dispatch(stream, [=](hipStream_t s) {
        dim3 grid((unsigned)cdiv(cols, QE_BT), (unsigned)std::min<int64_t>(rows, 65535));
        PMT_LAUNCH(bilinear_kernel, grid, dim3(256), 0, s, Q, ldq, rows, cols, xvar, yvar, moi, varmap, reinterpret_cast<u64 *>(out_quad));
        return check_launch("bilinear_kernel");
    }, nd);
}

extern "C" int pmt_vecdot_terms_f64(int64_t n, const double *xc, const int64_t *xvar, const double *yc, const int64_t *yvar, int moi,
                                    const int64_t *varmap, pmt_quadratic_term *out_quad, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "vecdot_terms: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(xvar && yvar && out_quad, PMT_INVALID_ARGUMENT, "vecdot_terms: null pointer");
    SmallNode nd;
    nd.op = SOP_VECDOT_TERMS; nd.moi = moi; nd.d[0] = n; nd.in[0] = xc; nd.in[1] = xvar; nd.in[2] = yc; nd.in[3] = yvar; nd.in[4] = varmap;
    nd.out[0] = out_quad; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(vecdot_terms_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, n, xc, xvar, yc, yvar, moi, varmap, out_quad);
        return check_launch("vecdot_terms_kernel");
    }, nd);
}

extern "C" int pmt_vecdot_affs_vars_f64(int64_t rows, const pmt_linear_term *x_terms, int64_t L, const double *x_consts, const int64_t *yvar,
                                        int moi, const int64_t *varmap, pmt_quadratic_term *out_quad, pmt_linear_term *out_lin,
                                        void *stream) {
    PMT_REQUIRE(rows >= 0 && L >= 0, PMT_DIMENSION_MISMATCH, "vecdot_affs_vars: negative dimension");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE(x_consts && yvar && out_lin && (L == 0 || (x_terms && out_quad)), PMT_INVALID_ARGUMENT, "vecdot_affs_vars: null pointer");
    SmallNode nd;
    nd.op = SOP_VECDOT_AFFS_VARS; nd.moi = moi; nd.d[0] = rows; nd.d[1] = L; nd.in[0] = x_terms; nd.in[1] = x_consts; nd.in[2] = yvar; nd.in[3] = varmap;
    nd.out[0] = out_quad; nd.out[1] = out_lin; nd.work = rows * L + rows;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(rows * L + rows, 256), 256 * 8);
        PMT_LAUNCH(vecdot_affs_vars_kernel, dim3(blocks), dim3(256), 0, s, rows, x_terms, L, x_consts, yvar, moi, varmap, out_quad,
                           out_lin);
        return check_launch("vecdot_affs_vars_kernel");
    }, nd);
}
