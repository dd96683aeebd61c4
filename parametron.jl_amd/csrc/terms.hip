// Generic term-list nodes on materialised Vector{AffineFunction} blocks (flat LinearTerm buffer + row_ptr +
// constants).  These cover the builders that are not on BASELINE's dense fast path; they are simple
// coalesced streaming kernels (one wave per row, or one thread per term).
//
// Reference loops replaced (see include/parametron_hip.h):
//   copyto!/add!/subtract! on AffineFunction   src/functions.jl:422-427,455,477-485 (vecadd!/vecsubtract! :751-764, vcat! :969-994)
//   scale!/mul!/muladd!(aff, number)           src/functions.jl:895-915,578,515-523
//   matvecmul!(y, A, x::Vector{AffineFunction}) src/functions.jl:800-822
//   _vecdot! affine forms                      src/functions.jl:665-687
#include "common.h"

namespace pmt {

int launch_seq_dot(const double *a, int sign_a, const double *b, int sign_b, int64_t n, double *out, hipStream_t s);

__global__ __launch_bounds__(256) void affvec_combine_kernel(
    int64_t rows,
    const LT *__restrict__ xa, const int64_t *__restrict__ xa_ptr, int64_t xa_len, const double *__restrict__ ca,
    const LT *__restrict__ xb, const int64_t *__restrict__ xb_ptr, int64_t xb_len, const double *__restrict__ cb, int sb,
    LT *__restrict__ out, const int64_t *__restrict__ out_ptr, int64_t out_len, double *__restrict__ out_consts) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t obeg = out_ptr ? out_ptr[row] : row * out_len;
    int64_t na = 0;
    if (xa) {
        const int64_t beg = xa_ptr ? xa_ptr[row] : row * xa_len;
        na = (xa_ptr ? xa_ptr[row + 1] : beg + xa_len) - beg;
        for (int64_t k = lane; k < na; k += 64) out[obeg + k] = xa[beg + k];
    }
    if (xb) {
        const int64_t beg = xb_ptr ? xb_ptr[row] : row * xb_len;
        const int64_t nb = (xb_ptr ? xb_ptr[row + 1] : beg + xb_len) - beg;
        for (int64_t k = lane; k < nb; k += 64) {
            LT t = xb[beg + k];
            if (sb < 0) t.coeff = -t.coeff;                       // -x.linear[i]  (functions.jl:481, :158)
            out[obeg + na + k] = t;
        }
    }
    if (lane == 0 && out_consts) {
        double c = ca ? ca[row] : 0.0;                            // copyto! :425 / :419 / zero! :244
        if (cb) c = sb < 0 ? c - cb[row] : c + cb[row];           // :483 / :455 / :452 / :474
        out_consts[row] = c;
    }
}

__global__ void affvec_scale_kernel(int64_t rows, int64_t nterms, const LT *__restrict__ y, const double *__restrict__ yc,
                                    const double *__restrict__ s_dev, double s_host, LT *__restrict__ out, double *__restrict__ out_consts) {
    const double s = s_dev ? *s_dev : s_host;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nterms + rows; i += stride) {
        if (i < nterms) {
            LT t = y[i];
            t.coeff = s * t.coeff;                                // x.linear[i] * y -> y * coeff (functions.jl:519, :159-160)
            out[i] = t;
        } else {
            const int64_t r = i - nterms;
            out_consts[r] = 0.0 + yc[r] * s;                      // zero! then dest.constant += x.constant * y (:521)
        }
    }
}

__global__ void matvecmul_affs_kernel(const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                      const LT *__restrict__ x, int64_t L, LT *__restrict__ out) {
    const int64_t per_row = cols * L;
    const int64_t total = rows * per_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t row = idx / per_row;
        const int64_t rem = idx - row * per_row;
        const int64_t col = rem / L;
        LT t = x[rem];                                             // x[col].linear[k], rem == col*L + k
        t.coeff = A[col * lda + row] * t.coeff;                    // muladd!(y[row], A[i], x[col]) (:817 -> :519)
        out[idx] = t;
    }
}
// const[row] = ((0 + xc[0]*A[row,0]) + xc[1]*A[row,1]) + ...   (functions.jl:521 in column order :815-820)
__global__ void matvecmul_affs_consts_kernel(const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                             const double *__restrict__ xc, double *__restrict__ out_consts) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double acc = 0.0;
    for (int64_t col = 0; col < cols; ++col) {
        const double p = xc[col] * A[col * lda + row];
        acc = acc + p;
    }
    out_consts[row] = acc;
}

__global__ void vecdot_numbers_vars_kernel(const double *__restrict__ v, const int64_t *__restrict__ xvar, int64_t n,
                                           LT *__restrict__ out, double *__restrict__ out_const) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && out_const) *out_const = 0.0;
    if (i >= n) return;
    LT t; t.coeff = v[i]; t.var = xvar[i];                         // x[i] * y[i] (functions.jl:684, :120)
    out[i] = t;
}

__global__ void vecdot_numbers_affs_kernel(const double *__restrict__ v, int64_t n, const LT *__restrict__ x, int64_t L,
                                           LT *__restrict__ out) {
    const int64_t total = n * L;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t i = idx / L;
        LT t = x[idx];
        t.coeff = v[i] * t.coeff;                                  // muladd!(dest, x::Number, y::Aff) (:524 -> :519)
        out[idx] = t;
    }
}

// dest[j, i] = A[i, j]: the `adjoint` rewrite rule's closure (src/lazyexpression.jl:206-217); 32x32 LDS tile transpose
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ src, int64_t lds_, int64_t rows, int64_t cols,
                                                        double *__restrict__ dst, int64_t ldd) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + tx, c = c0 + k;
        if (r < rows && c < cols) tile[k][tx] = src[c * lds_ + r];    // column-major read, rows fastest
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + tx, r = r0 + k;
        if (r < rows && c < cols) dst[r * ldd + c] = tile[tx][k];     // dst is cols x rows column-major: dst[c, r]
    }
}

// quadratic term lists: out = [ qa ; sb * qb ]  (copyto! :434-439, add! :459, subtract! :492-500) or s * q (muladd! :526-534)
__global__ void quad_combine_kernel(const QT *__restrict__ qa, int64_t na, const QT *__restrict__ qb, int64_t nb, int sb, QT *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
        QT t = i < na ? qa[i] : qb[i - na];
        if (i >= na && sb < 0) t.coeff = -t.coeff;
        out[i] = t;
    }
}
__global__ void quad_scale_kernel(const QT *__restrict__ q, int64_t n, const double *__restrict__ s_dev, double s_host, QT *__restrict__ out) {
    const double s = s_dev ? *s_dev : s_host;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        QT t = q[i];
        t.coeff = s * t.coeff;                                          // x.quadratic[i] * y -> y * coeff (functions.jl:530, :159-160)
        out[i] = t;
    }
}
// scale!(dest::Vector{LinearTerm}, x::Number, y::Vector{Variable}) src/functions.jl:873-893: dest[i] = (s, yvar[i])
__global__ void scale_vars_kernel(const int64_t *__restrict__ yvar, int64_t n, const double *__restrict__ s_dev, double s_host, LT *__restrict__ out) {
    const double s = s_dev ? *s_dev : s_host;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { LT t; t.coeff = s; t.var = yvar[i]; out[i] = t; }
}
// scale!(dest::Array{<:Number}, x::Number, y::Array{<:Number}) src/functions.jl:917-925: dest .= x .* y
__global__ void scale_numbers_kernel(const double *__restrict__ y, int64_t n, const double *__restrict__ s_dev, double s_host, double *__restrict__ out) {
    const double s = s_dev ? *s_dev : s_host;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = s * y[i];
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_transpose_f64(const double *src, int64_t lds_, int64_t rows, int64_t cols, double *dst, int64_t ldd, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "transpose: negative dimension");
    PMT_REQUIRE(lds_ >= rows && ldd >= cols, PMT_DIMENSION_MISMATCH, "transpose: leading dimension too small");
    if (rows == 0 || cols == 0) return PMT_OK;
    PMT_REQUIRE(src && dst, PMT_INVALID_ARGUMENT, "transpose: null pointer");
    SmallNode nd;
    nd.op = SOP_TRANSPOSE; nd.d[0] = lds_; nd.d[1] = rows; nd.d[2] = cols; nd.d[3] = ldd; nd.in[0] = src; nd.out[0] = dst; nd.work = rows * cols;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(transpose_kernel, dim3((unsigned)cdiv(rows, 32), (unsigned)cdiv(cols, 32)), dim3(256), 0, s, src, lds_, rows, cols, dst, ldd);
        return check_launch("transpose_kernel");
    }, nd);
}

extern "C" int pmt_quad_combine_f64(const pmt_quadratic_term *qa, int64_t na, const pmt_quadratic_term *qb, int64_t nb, int sb,
                                    pmt_quadratic_term *out, void *stream) {
    PMT_REQUIRE(na >= 0 && nb >= 0, PMT_DIMENSION_MISMATCH, "quad_combine: negative length");
    PMT_REQUIRE(sb == 1 || sb == -1, PMT_INVALID_ARGUMENT, "quad_combine: sb must be +1 or -1");
    if (na + nb == 0) return PMT_OK;
    PMT_REQUIRE(out && (na == 0 || qa) && (nb == 0 || qb), PMT_INVALID_ARGUMENT, "quad_combine: null pointer");
    SmallNode nd;
    nd.op = SOP_QUAD_COMBINE; nd.sign = sb; nd.d[0] = na; nd.d[1] = nb; nd.in[0] = qa; nd.in[1] = qb; nd.out[0] = out; nd.work = na + nb;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(na + nb, 256), 256 * 8);
        PMT_LAUNCH(quad_combine_kernel, dim3(blocks), dim3(256), 0, s, qa, na, qb, nb, sb, out);
        return check_launch("quad_combine_kernel");
    }, nd);
}

extern "C" int pmt_quad_scale_f64(const pmt_quadratic_term *q, int64_t n, const double *s_dev, double s_host, pmt_quadratic_term *out,
                                  void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "quad_scale: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(q && out, PMT_INVALID_ARGUMENT, "quad_scale: null pointer");
    SmallNode nd;
    nd.op = SOP_QUAD_SCALE; nd.d[0] = n; nd.in[0] = q; nd.in[1] = s_dev; nd.scale = s_host; nd.out[0] = out; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 256), 256 * 8);
        PMT_LAUNCH(quad_scale_kernel, dim3(blocks), dim3(256), 0, s, q, n, s_dev, s_host, out);
        return check_launch("quad_scale_kernel");
    }, nd);
}

extern "C" int pmt_scale_vars_f64(const int64_t *yvar, int64_t n, const double *s_dev, double s_host, pmt_linear_term *out, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "scale_vars: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(yvar && out, PMT_INVALID_ARGUMENT, "scale_vars: null pointer");
    SmallNode nd;
    nd.op = SOP_SCALE_VARS; nd.d[0] = n; nd.in[0] = yvar; nd.in[1] = s_dev; nd.scale = s_host; nd.out[0] = out; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(scale_vars_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, yvar, n, s_dev, s_host, out);
        return check_launch("scale_vars_kernel");
    }, nd);
}

extern "C" int pmt_scale_numbers_f64(const double *y, int64_t n, const double *s_dev, double s_host, double *out, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "scale_numbers: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(y && out, PMT_INVALID_ARGUMENT, "scale_numbers: null pointer");
    SmallNode nd;
    nd.op = SOP_SCALE_NUMBERS; nd.d[0] = n; nd.in[0] = y; nd.in[1] = s_dev; nd.scale = s_host; nd.out[0] = out; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 256), 256 * 8);
        PMT_LAUNCH(scale_numbers_kernel, dim3(blocks), dim3(256), 0, s, y, n, s_dev, s_host, out);
        return check_launch("scale_numbers_kernel");
    }, nd);
}

extern "C" int pmt_copy_bytes(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return PMT_OK;
    PMT_REQUIRE(dst && src, PMT_INVALID_ARGUMENT, "copy_bytes: null pointer");
    Launch copy = [=](hipStream_t s) {
        // (hipMemcpyDefault: `src` may be page-locked HOST memory — a small model's Parameter mailbox, which the small-plan kernel reads directly)
        PMT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
        return PMT_OK;
    };
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 7) == 0) {
        SmallNode nd;
        nd.op = SOP_COPY8; nd.d[0] = (int64_t)(bytes / 8); nd.in[0] = src; nd.out[0] = dst; nd.work = (int64_t)(bytes / 8);
        return dispatch(stream, copy, nd);
    }
    return dispatch(stream, copy);
}

extern "C" int pmt_affvec_combine_f64(int64_t rows, const pmt_linear_term *xa_terms, const int64_t *xa_row_ptr, int64_t xa_row_len,
                                      const double *xa_consts, const pmt_linear_term *xb_terms, const int64_t *xb_row_ptr,
                                      int64_t xb_row_len, const double *xb_consts, int sb, pmt_linear_term *out_terms,
                                      const int64_t *out_row_ptr, int64_t out_row_len, double *out_consts, void *stream) {
    PMT_REQUIRE(rows >= 0 && xa_row_len >= 0 && xb_row_len >= 0 && out_row_len >= 0, PMT_DIMENSION_MISMATCH, "affvec_combine: negative dimension");
    PMT_REQUIRE(sb == 1 || sb == -1, PMT_INVALID_ARGUMENT, "affvec_combine: sb must be +1 or -1");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE(out_consts, PMT_INVALID_ARGUMENT, "affvec_combine: null out_consts");
    if (!xa_row_ptr && !xb_row_ptr && !out_row_ptr)
        PMT_REQUIRE((xa_terms ? xa_row_len : 0) + (xb_terms ? xb_row_len : 0) == out_row_len, PMT_DIMENSION_MISMATCH,
                    "affvec_combine: out_row_len != len(a) + len(b)");
    PMT_REQUIRE(out_terms || out_row_len == 0, PMT_INVALID_ARGUMENT, "affvec_combine: null out_terms");
    SmallNode nd;
    nd.op = SOP_AFFVEC_COMBINE; nd.sign = sb; nd.d[0] = rows; nd.d[1] = xa_row_len; nd.d[2] = xb_row_len; nd.d[3] = out_row_len;
    nd.in[0] = xa_terms; nd.in[1] = xa_consts; nd.in[2] = xb_terms; nd.in[3] = xb_consts; nd.out[0] = out_terms; nd.out[1] = out_consts;
    // (ragged rows — any row_ptr — keep their own kernel: the term count lives on the device)
    nd.work = (xa_row_ptr || xb_row_ptr || out_row_ptr) ? SMALL_NODE_WORK_MAX + 1 : rows * out_row_len + rows;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(affvec_combine_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, rows, xa_terms, xa_row_ptr, xa_row_len,
                           xa_consts, xb_terms, xb_row_ptr, xb_row_len, xb_consts, sb, out_terms, out_row_ptr, out_row_len, out_consts);
        return check_launch("affvec_combine_kernel");
    }, nd);
}

extern "C" int pmt_affvec_scale_f64(int64_t rows, int64_t nterms, const pmt_linear_term *y_terms, const double *y_consts, const double *s_dev,
                                    double s_host, pmt_linear_term *out_terms, double *out_consts, void *stream) {
    PMT_REQUIRE(rows >= 0 && nterms >= 0, PMT_DIMENSION_MISMATCH, "affvec_scale: negative dimension");
    if (rows == 0 && nterms == 0) return PMT_OK;
    PMT_REQUIRE((nterms == 0 || (y_terms && out_terms)) && (rows == 0 || (y_consts && out_consts)), PMT_INVALID_ARGUMENT, "affvec_scale: null pointer");
    SmallNode nd;
    nd.op = SOP_AFFVEC_SCALE; nd.d[0] = rows; nd.d[1] = nterms; nd.in[0] = y_terms; nd.in[1] = y_consts; nd.in[2] = s_dev; nd.scale = s_host;
    nd.out[0] = out_terms; nd.out[1] = out_consts; nd.work = nterms + rows;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(nterms + rows, 256), 256 * 8);
        PMT_LAUNCH(affvec_scale_kernel, dim3(blocks), dim3(256), 0, s, rows, nterms, y_terms, y_consts, s_dev, s_host, out_terms, out_consts);
        return check_launch("affvec_scale_kernel");
    }, nd);
}

extern "C" int pmt_matvecmul_affs_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const pmt_linear_term *x_terms,
                                      int64_t x_row_len, const double *x_consts, pmt_linear_term *out_terms, double *out_consts,
                                      void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0 && x_row_len >= 0, PMT_DIMENSION_MISMATCH, "matvecmul_affs: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "matvecmul_affs: lda < rows");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE(out_consts && (cols == 0 || (A && x_consts)), PMT_INVALID_ARGUMENT, "matvecmul_affs: null pointer");
    PMT_REQUIRE(cols * x_row_len == 0 || (x_terms && out_terms), PMT_INVALID_ARGUMENT, "matvecmul_affs: null terms");
    SmallNode nd;
    nd.op = SOP_MATVEC_AFFS; nd.d[0] = lda; nd.d[1] = rows; nd.d[2] = cols; nd.d[3] = x_row_len; nd.in[0] = A; nd.in[1] = x_terms; nd.in[2] = x_consts;
    nd.out[0] = out_terms; nd.out[1] = out_consts; nd.work = rows * cols * x_row_len + rows * cols;
    return dispatch(stream, [=](hipStream_t s) {
        if (cols * x_row_len > 0) {
            const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(rows * cols * x_row_len, 256), 256 * 16);
            PMT_LAUNCH(matvecmul_affs_kernel, dim3(blocks), dim3(256), 0, s, A, lda, rows, cols, x_terms, x_row_len, out_terms);
            int rc = check_launch("matvecmul_affs_kernel");
            if (rc) return rc;
        }
        PMT_LAUNCH(matvecmul_affs_consts_kernel, dim3((unsigned)cdiv(rows, 64)), dim3(64), 0, s, A, lda, rows, cols, x_consts, out_consts);
        return check_launch("matvecmul_affs_consts_kernel");
    }, nd);
}

extern "C" int pmt_vecdot_numbers_vars_f64(const double *v, const int64_t *xvar, int64_t n, pmt_linear_term *out_terms, double *out_const,
                                           void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "vecdot_numbers_vars: negative length");
    PMT_REQUIRE(out_const && (n == 0 || (v && xvar && out_terms)), PMT_INVALID_ARGUMENT, "vecdot_numbers_vars: null pointer");
    SmallNode nd;
    nd.op = SOP_VECDOT_NUM_VARS; nd.d[0] = n; nd.in[0] = v; nd.in[1] = xvar; nd.out[0] = out_terms; nd.out[1] = out_const; nd.work = n + 1;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(vecdot_numbers_vars_kernel, dim3((unsigned)std::max<int64_t>(1, cdiv(n, 256))), dim3(256), 0, s, v, xvar, n, out_terms, out_const);
        return check_launch("vecdot_numbers_vars_kernel");
    }, nd);
}

extern "C" int pmt_vecdot_numbers_affs_f64(const double *v, int64_t n, const pmt_linear_term *x_terms, int64_t x_row_len, const double *x_consts,
                                           pmt_linear_term *out_terms, double *out_const, void *stream) {
    PMT_REQUIRE(n >= 0 && x_row_len >= 0, PMT_DIMENSION_MISMATCH, "vecdot_numbers_affs: negative dimension");
    PMT_REQUIRE(out_const && (n == 0 || (v && x_consts)), PMT_INVALID_ARGUMENT, "vecdot_numbers_affs: null pointer");
    PMT_REQUIRE(n * x_row_len == 0 || (x_terms && out_terms), PMT_INVALID_ARGUMENT, "vecdot_numbers_affs: null terms");
    SmallNode nd;
    nd.op = SOP_VECDOT_NUM_AFFS; nd.d[0] = n; nd.d[1] = x_row_len; nd.in[0] = v; nd.in[1] = x_terms; nd.in[2] = x_consts; nd.out[0] = out_terms;
    nd.out[1] = out_const; nd.work = n > 2048 ? SMALL_NODE_WORK_MAX + 1 : n * x_row_len + 16 * n;      // (its constant: a chain on one thread)
    return dispatch(stream, [=](hipStream_t s) {
        if (n * x_row_len > 0) {
            const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n * x_row_len, 256), 256 * 8);
            PMT_LAUNCH(vecdot_numbers_affs_kernel, dim3(blocks), dim3(256), 0, s, v, n, x_terms, x_row_len, out_terms);
            int rc = check_launch("vecdot_numbers_affs_kernel");
            if (rc) return rc;
        }
        return launch_seq_dot(x_consts, 2, v, 2, n, out_const, s);   // dest.constant += x.constant * y (functions.jl:521)
    }, nd);
}
