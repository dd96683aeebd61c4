// Solver hand-off: the step AFTER MOI.set(optimizer, ...) (src/moi_interop.jl:134,171), which in the
// reference is third-party code (MathOptInterface 0.8 + the OSQP wrapper): turn the MOI functions into the solver's matrices
//     minimize 1/2 x'Px + q'x   subject to   l <= Ax <= u          (P upper triangular, P and A in CSC, 0-based Int64)
// Here the MOI buffers stay in HBM and the CSC VALUES are rebuilt on the device at every re-evaluation; the CSC STRUCTURE
// depends only on the (static) indices and is computed once on the host (pmt_csc_order), like the sort of canonicalize!.
//
// Semantics restated (MathOptInterface 0.8 ScalarQuadraticFunction docstring: the function is 1/2 x'Qx + a'x + c with Q
// symmetric; a term (c, i, j), i != j, stands for Q_ij = Q_ji = c; duplicates add): P_ij = sum of the coefficients of the terms
// on {i, j} — the MOI coefficients are used as they are, the diagonal doubling already happened at moi_interop.jl:58.
// A row of `f(x) in set` with f = a'x + c becomes  a'x in [l, u]:  Zeros/EqualTo(v): l = u = v - c;  Nonnegatives/GreaterThan(v):
// l = v - c, u = +infty;  Nonpositives/LessThan(v): l = -infty, u = v - c.
#include <algorithm>
#include <numeric>
#include <new>
#include <vector>

#include "common.h"

namespace pmt {

// dst[dst_index ? dst_index[s] : s] = alpha * (sum of the run s), added in the original order of the terms (deterministic).
// One THREAD per run: the runs of a solver matrix are short (1 for canonical objectives and dense constraint blocks).
__global__ __launch_bounds__(256) void csc_values_thread_kernel(const char *__restrict__ src, int64_t stride, const int64_t *__restrict__ perm,
                                                                const int64_t *__restrict__ seg_ptr, int64_t nseg, double alpha,
                                                                const int64_t *__restrict__ dst_index, double *__restrict__ dst) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const int64_t p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    double acc = *reinterpret_cast<const double *>(src + perm[p0] * stride);
    for (int64_t p = p0 + 1; p < p1; ++p) acc += *reinterpret_cast<const double *>(src + perm[p] * stride);
    dst[dst_index ? dst_index[s] : s] = alpha * acc;
}

// The same with one ABSOLUTE coefficient address per term (several term buffers feeding one matrix in a single launch)
__global__ __launch_bounds__(256) void csc_values_gather_kernel(const double *const *__restrict__ term_ptr, const int64_t *__restrict__ seg_ptr,
                                                                int64_t nseg, double alpha, const int64_t *__restrict__ dst_index,
                                                                double *__restrict__ dst) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const int64_t p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    double acc = *term_ptr[p0];
    for (int64_t p = p0 + 1; p < p1; ++p) acc += *term_ptr[p];
    dst[dst_index ? dst_index[s] : s] = alpha * acc;
}

// One WAVE per run for long runs (literal r*n^2 objectives: every (j,k) has r duplicates): lanes stride the run, fixed butterfly.
__global__ __launch_bounds__(256) void csc_values_wave_kernel(const char *__restrict__ src, int64_t stride, const int64_t *__restrict__ perm,
                                                              const int64_t *__restrict__ seg_ptr, int64_t nseg, double alpha,
                                                              const int64_t *__restrict__ dst_index, double *__restrict__ dst) {
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    double acc = 0.0;
    for (int64_t p = p0 + lane; p < p1; p += 64) acc += *reinterpret_cast<const double *>(src + perm[p] * stride);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) dst[dst_index ? dst_index[s] : s] = alpha * acc;
}

// The CSC values of a DENSE constraint block are its Parameter matrix column by column (update!(::MOI.VectorAffineFunction) writes one term
// per entry, src/moi_interop.jl:64-81): column j of the block = `rows` consecutive doubles at src + j * spitch, going to the block's row range
// of column j of the stacked matrix, dst + j * dpitch.  One wave per column, lanes along it (coalesced both ways); 16.8 MB read for 16.8 MB
// written at config 2, where the per-term pointer gather this replaces read 248 MB (profiles/r03_pmc_traffic.json).
// dst_offset != null: column j goes to dst + dst_offset[j] (the other blocks of the stacked matrix are not the same height in every column).
__global__ __launch_bounds__(256) void copy_2d_kernel(const double *__restrict__ src, int64_t spitch, double *__restrict__ dst, int64_t dpitch,
                                                      const int64_t *__restrict__ dst_offset, int rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t j = wave0; j < cols; j += nwaves) {
        const double *s = src + j * spitch;
        double *d = dst + (dst_offset ? dst_offset[j] : j * dpitch);
        for (int i = lane; i < rows; i += 64) d[i] = s[i];
    }
}

__global__ __launch_bounds__(256) void qp_bounds_kernel(const double *__restrict__ consts, int64_t rows, int kind, double value, double infty,
                                                        double *__restrict__ l, double *__restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double b = value - consts[i];
    l[i] = kind == PMT_SET_LESS ? -infty : b;
    u[i] = kind == PMT_SET_GREATER ? infty : b;
}

// rows of several constraint blocks in one launch: row i reads its constant through const_ptr[i]
__global__ __launch_bounds__(256) void qp_bounds_rows_kernel(const double *const *__restrict__ const_ptr, const int *__restrict__ kind,
                                                             const double *__restrict__ value, int64_t rows, double infty,
                                                             double *__restrict__ l, double *__restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double b = value[i] - *const_ptr[i];
    const int k = kind[i];
    l[i] = k == PMT_SET_LESS ? -infty : b;
    u[i] = k == PMT_SET_GREATER ? infty : b;
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_csc_order(int64_t nnz_in, const int64_t *rows, const int64_t *cols, int64_t nrows, int64_t ncols, int upper,
                             int64_t *perm, int64_t *seg_ptr, int64_t *col_ptr, int64_t *row_idx, int64_t *nnz_out) try {
    PMT_REQUIRE(nnz_in >= 0 && nrows >= 0 && ncols >= 0, PMT_DIMENSION_MISMATCH, "csc_order: negative size");
    PMT_REQUIRE(nnz_out && seg_ptr && col_ptr && (nnz_in == 0 || (rows && cols && perm && row_idx)), PMT_INVALID_ARGUMENT, "csc_order: null pointer");
    auto r_of = [&](int64_t i) { return (upper ? std::min(rows[i], cols[i]) : rows[i]) - 1; };
    auto c_of = [&](int64_t i) { return (upper ? std::max(rows[i], cols[i]) : cols[i]) - 1; };
    for (int64_t i = 0; i < nnz_in; ++i)
        PMT_REQUIRE(r_of(i) >= 0 && r_of(i) < nrows && c_of(i) >= 0 && c_of(i) < ncols, PMT_DIMENSION_MISMATCH,
                    "csc_order: index out of range (indices are 1-based)");
    // counting sort by column (stable), then a stable sort by row inside each column
    std::vector<int64_t> start((size_t)ncols + 1, 0);
    for (int64_t i = 0; i < nnz_in; ++i) ++start[(size_t)c_of(i) + 1];
    for (int64_t c = 0; c < ncols; ++c) start[(size_t)c + 1] += start[(size_t)c];
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < nnz_in; ++i) perm[fill[(size_t)c_of(i)]++] = i;
    }
    int64_t nseg = 0;
    for (int64_t c = 0; c < ncols; ++c) {
        int64_t *b = perm + start[(size_t)c], *e = perm + start[(size_t)c + 1];
        std::stable_sort(b, e, [&](int64_t x, int64_t y) { return r_of(x) < r_of(y); });
        col_ptr[c] = nseg;
        for (int64_t *p = b; p < e; ++p) {
            if (p == b || r_of(p[-1]) != r_of(*p)) {
                seg_ptr[nseg] = p - perm;
                row_idx[nseg] = r_of(*p);
                ++nseg;
            }
        }
    }
    col_ptr[ncols] = nseg;
    seg_ptr[nseg] = nnz_in;
    *nnz_out = nseg;
    return PMT_OK;
} catch (const std::bad_alloc &) {
    return pmt::fail(PMT_OUT_OF_MEMORY, "csc_order: out of host memory");
}

extern "C" int pmt_csc_values_f64(const void *src_coeff, int64_t src_stride_bytes, int64_t nnz_in, const int64_t *perm, const int64_t *seg_ptr,
                                  int64_t nseg, double alpha, const int64_t *dst_index, double *dst_values, void *stream) {
    PMT_REQUIRE(nseg >= 0 && nnz_in >= nseg, PMT_DIMENSION_MISMATCH, "csc_values: need 0 <= nseg <= nnz_in");
    PMT_REQUIRE(src_stride_bytes >= 8 && (src_stride_bytes % 8) == 0, PMT_INVALID_ARGUMENT, "csc_values: stride must be a multiple of 8 bytes");
    if (nseg == 0) return PMT_OK;
    PMT_REQUIRE(src_coeff && perm && seg_ptr && dst_values, PMT_INVALID_ARGUMENT, "csc_values: null pointer");
    const bool long_runs = nnz_in >= 32 * nseg;
    return dispatch(stream, [=](hipStream_t s) {
        const char *src = reinterpret_cast<const char *>(src_coeff);
        if (long_runs) {
            PMT_LAUNCH(csc_values_wave_kernel, dim3((unsigned)cdiv(nseg, 4)), dim3(256), 0, s, src, src_stride_bytes, perm, seg_ptr, nseg, alpha,
                       dst_index, dst_values);
            return check_launch("csc_values_wave_kernel");
        }
        PMT_LAUNCH(csc_values_thread_kernel, dim3((unsigned)cdiv(nseg, 256)), dim3(256), 0, s, src, src_stride_bytes, perm, seg_ptr, nseg, alpha,
                   dst_index, dst_values);
        return check_launch("csc_values_thread_kernel");
    });
}

extern "C" int pmt_qp_bounds_f64(const double *consts, int64_t rows, int set_kind, double set_value, double infty, double *l, double *u,
                                 void *stream) {
    PMT_REQUIRE(rows >= 0, PMT_DIMENSION_MISMATCH, "qp_bounds: negative row count");
    PMT_REQUIRE(set_kind == PMT_SET_EQUAL || set_kind == PMT_SET_GREATER || set_kind == PMT_SET_LESS, PMT_INVALID_ARGUMENT, "qp_bounds: unknown set kind");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE(consts && l && u, PMT_INVALID_ARGUMENT, "qp_bounds: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(qp_bounds_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0, s, consts, rows, set_kind, set_value, infty, l, u);
        return check_launch("qp_bounds_kernel");
    });
}

extern "C" int pmt_csc_values_gather_f64(const double *const *term_ptr, int64_t nnz_in, const int64_t *seg_ptr, int64_t nseg, double alpha,
                                         const int64_t *dst_index, double *dst_values, void *stream) {
    PMT_REQUIRE(nseg >= 0 && nnz_in >= nseg, PMT_DIMENSION_MISMATCH, "csc_values_gather: need 0 <= nseg <= nnz_in");
    if (nseg == 0) return PMT_OK;
    PMT_REQUIRE(term_ptr && seg_ptr && dst_values, PMT_INVALID_ARGUMENT, "csc_values_gather: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(csc_values_gather_kernel, dim3((unsigned)cdiv(nseg, 256)), dim3(256), 0, s, term_ptr, seg_ptr, nseg, alpha, dst_index, dst_values);
        return check_launch("csc_values_gather_kernel");
    });
}

extern "C" int pmt_copy_2d_f64(const double *src, int64_t src_pitch, double *dst, int64_t dst_pitch, const int64_t *dst_offset, int64_t rows,
                               int64_t cols, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0 && rows < ((int64_t)1 << 31), PMT_DIMENSION_MISMATCH, "copy_2d: bad size");
    PMT_REQUIRE(src_pitch >= rows && (dst_offset || dst_pitch >= rows), PMT_DIMENSION_MISMATCH, "copy_2d: pitch smaller than the column");
    if (rows == 0 || cols == 0) return PMT_OK;
    PMT_REQUIRE(src && dst, PMT_INVALID_ARGUMENT, "copy_2d: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(copy_2d_kernel, dim3((unsigned)std::min<int64_t>(cdiv(cols, 4), 4096)), dim3(256), 0, s, src, src_pitch, dst, dst_pitch, dst_offset, (int)rows, cols);
        return check_launch("copy_2d_kernel");
    });
}

extern "C" int pmt_qp_bounds_rows_f64(const double *const *const_ptr, const int *set_kind, const double *set_value, int64_t rows, double infty,
                                      double *l, double *u, void *stream) {
    PMT_REQUIRE(rows >= 0, PMT_DIMENSION_MISMATCH, "qp_bounds_rows: negative row count");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE(const_ptr && set_kind && set_value && l && u, PMT_INVALID_ARGUMENT, "qp_bounds_rows: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(qp_bounds_rows_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0, s, const_ptr, set_kind, set_value, rows, infty, l, u);
        return check_launch("qp_bounds_rows_kernel");
    });
}
