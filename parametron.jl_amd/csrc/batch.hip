// Batched independent QPs (BASELINE config 4: 8192 instances of n = r = 128, sharded by instance across GPUs).
// In the reference a batch is simply many Models (src/model.jl:1-22); nothing is shared between them.  Every instance has the
// same structure, hence the same index arrays, so per re-evaluation only COEFFICIENTS are produced (and exchanged between GPUs):
// one slab of L doubles per instance, laid out as
//     [ Q: n(n+1)/2 | q: n | const: 1 | C: m*n row-major | d-consts: m ]
// where Q/q/const are the canonical objective coefficients of residual . residual (MOI form) and C / d-consts
// the coefficients / constants of the constraint block C*x (+|-) d (Appendix A.4).  pmt_batch_expand_f64 turns one instance's slab
// into the full MOI term buffers (coefficient + indices) — byte-identical to what pmt_quad_gram_f64 / pmt_affine_pack_vector_f64
// produce for that instance alone.
#include "common.h"

namespace pmt {

int launch_batch_gram(const double *A, int64_t lda, int64_t rows, int64_t cols, int64_t strideA, const double *b, int64_t strideb, int sign,
                      int64_t B, double *out_q, double *out_lin, double *out_const, int64_t out_stride, hipStream_t s);
int launch_batch_small(const double *A, int64_t lda, int64_t rows, int64_t cols, int64_t strideA, const double *b, int64_t strideb, int sign,
                       int64_t B, double *out, int64_t out_stride, const double *Cm, int64_t m, const double *d, int sign_d, hipStream_t s);

// out[inst][row*cols + col] = C_inst[row, col] (column-major in, row-major out); 32x32 LDS tiles
__global__ __launch_bounds__(256) void batch_transpose_kernel(const double *__restrict__ src, int64_t rows, int64_t cols, int64_t stride_src,
                                                              double *__restrict__ dst, int64_t stride_dst) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    const double *s = src + (int64_t)blockIdx.z * stride_src;
    double *d = dst + (int64_t)blockIdx.z * stride_dst;
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + tx, c = c0 + k;
        if (r < rows && c < cols) tile[k][tx] = s[c * rows + r];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + tx, r = r0 + k;
        if (r < rows && c < cols) d[r * cols + c] = tile[tx][k];
    }
}

__global__ void batch_consts_kernel(const double *__restrict__ d, int64_t m, int64_t B, int sign, double *__restrict__ out, int64_t out_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * m) return;
    const int64_t inst = i / m, r = i - inst * m;
    out[inst * out_stride + r] = signed_const(d[i], sign);
}

__global__ void batch_expand_kernel(const double *__restrict__ slab, int64_t n, int64_t m, const int64_t *__restrict__ xvar,
                                    const int64_t *__restrict__ varmap, QT *__restrict__ oq, LT *__restrict__ ol, double *__restrict__ oc,
                                    VAT *__restrict__ ov, double *__restrict__ ovc) {
    const int64_t nq = n * (n + 1) / 2;
    const int64_t total = nq + n + 1 + m * n + m;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const double v = slab[i];
        if (i < nq) {
            // invert pos = j*n - j(j-1)/2 + (k-j)
            int64_t j = (int64_t)((2.0 * n + 1.0 - sqrt((2.0 * n + 1.0) * (2.0 * n + 1.0) - 8.0 * (double)i)) * 0.5);
            if (j < 0) j = 0;
            if (j > n - 1) j = n - 1;
            while (j > 0 && (j * n - j * (j - 1) / 2) > i) --j;
            while (j + 1 < n && ((j + 1) * n - (j + 1) * j / 2) <= i) ++j;
            const int64_t k = j + (i - (j * n - j * (j - 1) / 2));
            QT t; t.coeff = v; t.row = map_var(varmap, xvar[j]); t.col = map_var(varmap, xvar[k]);
            oq[i] = t;
        } else if (i < nq + n) {
            const int64_t j = i - nq;
            LT t; t.coeff = v; t.var = map_var(varmap, xvar[j]);
            ol[j] = t;
        } else if (i == nq + n) {
            *oc = v;
        } else if (i < nq + n + 1 + m * n) {
            const int64_t e = i - (nq + n + 1);
            const int64_t row = e / n, col = e - row * n;
            VAT t; t.output_index = row + 1; t.coeff = v; t.var = map_var(varmap, xvar[col]);
            ov[e] = t;
        } else {
            ovc[i - (nq + n + 1 + m * n)] = v;
        }
    }
}

}  // namespace pmt

using namespace pmt;

extern "C" int64_t pmt_batch_lsq_slab_doubles(int64_t n, int64_t m) { return n * (n + 1) / 2 + n + 1 + m * n + m; }

extern "C" int pmt_batch_lsq_coeffs_f64(const double *A, const double *b, const double *Cm, const double *d, int64_t B, int64_t n, int64_t r,
                                        int64_t m, int sign_b, int sign_d, double *out, int64_t out_stride, void *stream) {
    PMT_REQUIRE(B >= 0 && n >= 0 && r >= 0 && m >= 0, PMT_DIMENSION_MISMATCH, "batch_lsq: negative dimension");
    const int64_t L = pmt_batch_lsq_slab_doubles(n, m);
    PMT_REQUIRE(out_stride >= L, PMT_DIMENSION_MISMATCH, "batch_lsq: out_stride smaller than the slab");
    PMT_REQUIRE(sign_b >= -1 && sign_b <= 1 && sign_d >= -1 && sign_d <= 1, PMT_INVALID_ARGUMENT, "batch_lsq: signs must be -1, 0 or +1");
    if (B == 0) return PMT_OK;
    PMT_REQUIRE(out && (n * r == 0 || A) && (r == 0 || b) && (m * n == 0 || Cm) && (m == 0 || d), PMT_INVALID_ARGUMENT, "batch_lsq: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        const int64_t nq = n * (n + 1) / 2;
        if (n > 0 && n <= 128) {
            // small instances: the whole slab of an instance from one persistent workgroup (batch_small.hip)
            return launch_batch_small(A, r, r, n, r * n, b, r, sign_b, B, out, out_stride, Cm, m, d, sign_d, s);
        }
        if (n > 0) {
            int rc = launch_batch_gram(A, r, r, n, r * n, b, r, sign_b, B, out, out + nq, out + nq + n, out_stride, s);
            if (rc) return rc;
        }
        if (m > 0 && n > 0) {
            for (int64_t i0 = 0; i0 < B; i0 += 65535) {
                const unsigned nb = (unsigned)std::min<int64_t>(65535, B - i0);
                PMT_LAUNCH(batch_transpose_kernel, dim3((unsigned)cdiv(m, 32), (unsigned)cdiv(n, 32), nb), dim3(256), 0, s, Cm + i0 * m * n, m, n,
                           m * n, out + i0 * out_stride + nq + n + 1, out_stride);
            }
        }
        if (m > 0) {
            PMT_LAUNCH(batch_consts_kernel, dim3((unsigned)cdiv(B * m, 256)), dim3(256), 0, s, d, m, B, sign_d, out + nq + n + 1 + m * n, out_stride);
        }
        return check_launch("batch_lsq");
    });
}

extern "C" int pmt_batch_expand_f64(const double *slab, int64_t n, int64_t m, const int64_t *xvar, const int64_t *varmap,
                                    pmt_quadratic_term *out_quad, pmt_linear_term *out_lin, double *out_const,
                                    pmt_vector_affine_term *out_vat, double *out_vconsts, void *stream) {
    PMT_REQUIRE(n >= 0 && m >= 0, PMT_DIMENSION_MISMATCH, "batch_expand: negative dimension");
    PMT_REQUIRE(slab && xvar && out_const && (n == 0 || (out_quad && out_lin)) && (m * n == 0 || out_vat) && (m == 0 || out_vconsts),
                PMT_INVALID_ARGUMENT, "batch_expand: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        const int64_t L = pmt_batch_lsq_slab_doubles(n, m);
        PMT_LAUNCH(batch_expand_kernel, dim3((unsigned)std::min<int64_t>(cdiv(L, 256), 2048)), dim3(256), 0, s, slab, n, m, xvar, varmap, out_quad,
                   out_lin, out_const, out_vat, out_vconsts);
        return check_launch("batch_expand_kernel");
    });
}
