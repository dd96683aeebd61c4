// Sparse constraint matrix (BASELINE config 5): C*x (+|-) d with C in CSC (Julia SparseMatrixCSC layout,
// 1-based Int64 colptr/rowval).  The reference has no sparse code path: matvecmul! (src/functions.jl:775-798)
// walks every (row, col) of an AbstractMatrix, so a sparse C yields explicit-zero terms.  This node emits the
// terms of the STRUCTURAL non-zeros only, in the reference's row-major order (row, then ascending column) —
// i.e. the reference output with the structural zeros removed.  The row-major permutation of the CSC
// non-zeros is computed once on the host at plan time (the sparsity pattern is fixed across re-evaluations);
// each re-evaluation is one gather + coalesced 24-byte AoS write: 8 (nzval) + 8 (perm) + 8 (row) + 8 (var) read,
// 24 written per non-zero.
#include <vector>

#include "common.h"

namespace pmt {

__global__ void sparse_pack_vector_kernel(const double *__restrict__ nzval, const int64_t *__restrict__ perm,
                                          const int64_t *__restrict__ term_row, const int64_t *__restrict__ term_var, int64_t nnz,
                                          const int64_t *__restrict__ varmap, int64_t row_offset, VAT *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += stride) {
        VAT o;
        o.output_index = row_offset + term_row[t];
        o.coeff = nzval[perm[t]];
        o.var = map_var(varmap, term_var[t]);
        out[t] = o;
    }
}

// native (LinearTerm) form of the same node: out[t] = (nzval[perm[t]], term_var[t]); rows are ragged (row_ptr from the plan)
__global__ void sparse_assemble_kernel(const double *__restrict__ nzval, const int64_t *__restrict__ perm,
                                       const int64_t *__restrict__ term_var, int64_t nnz, LT *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += stride) {
        LT o;
        o.coeff = nzval[perm[t]];
        o.var = term_var[t];
        out[t] = o;
    }
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_sparse_assemble_f64(const double *nzval, const int64_t *perm, const int64_t *term_var, int64_t nnz,
                                       pmt_linear_term *out_terms, void *stream) {
    PMT_REQUIRE(nnz >= 0, PMT_DIMENSION_MISMATCH, "sparse_assemble: negative nnz");
    if (nnz == 0) return PMT_OK;
    PMT_REQUIRE(nzval && perm && term_var && out_terms, PMT_INVALID_ARGUMENT, "sparse_assemble: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(nnz, 256), 256 * 8);
        PMT_LAUNCH(sparse_assemble_kernel, dim3(blocks), dim3(256), 0, s, nzval, perm, term_var, nnz, out_terms);
        return check_launch("sparse_assemble_kernel");
    });
}

extern "C" int pmt_sparse_rowmajor_order(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int64_t *perm,
                                         int64_t *rows_out, int64_t *cols_out, int64_t *row_ptr) {
    PMT_REQUIRE(m >= 0 && n >= 0, PMT_DIMENSION_MISMATCH, "sparse_rowmajor_order: negative dimension");
    PMT_REQUIRE(colptr && row_ptr, PMT_INVALID_ARGUMENT, "sparse_rowmajor_order: null pointer");
    PMT_REQUIRE(colptr[0] == 1, PMT_INVALID_ARGUMENT, "sparse_rowmajor_order: colptr must be 1-based");
    const int64_t nnz = colptr[n] - 1;
    PMT_REQUIRE(nnz == 0 || (rowval && perm && rows_out && cols_out), PMT_INVALID_ARGUMENT, "sparse_rowmajor_order: null pointer");
    for (int64_t i = 0; i <= m; ++i) row_ptr[i] = 0;
    for (int64_t c = 0; c < n; ++c) {
        PMT_REQUIRE(colptr[c + 1] >= colptr[c], PMT_INVALID_ARGUMENT, "sparse_rowmajor_order: colptr not monotone");
        for (int64_t p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p) {
            PMT_REQUIRE(rowval[p] >= 1 && rowval[p] <= m, PMT_DIMENSION_MISMATCH, "sparse_rowmajor_order: row index out of range");
            row_ptr[rowval[p]]++;
        }
    }
    for (int64_t i = 0; i < m; ++i) row_ptr[i + 1] += row_ptr[i];
    std::vector<int64_t> cursor(row_ptr, row_ptr + m);
    for (int64_t c = 0; c < n; ++c)               // ascending column => ascending column within each row
        for (int64_t p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p) {
            const int64_t r = rowval[p] - 1;
            const int64_t t = cursor[r]++;
            perm[t] = p;
            rows_out[t] = r + 1;
            cols_out[t] = c + 1;
        }
    return PMT_OK;
}

extern "C" int pmt_sparse_pack_vector_f64(const double *nzval, const int64_t *perm, const int64_t *term_row, const int64_t *term_var,
                                          int64_t nnz, const int64_t *varmap, int64_t row_offset, pmt_vector_affine_term *out_terms,
                                          void *stream) {
    PMT_REQUIRE(nnz >= 0, PMT_DIMENSION_MISMATCH, "sparse_pack_vector: negative nnz");
    if (nnz == 0) return PMT_OK;
    PMT_REQUIRE(nzval && perm && term_row && term_var && out_terms, PMT_INVALID_ARGUMENT, "sparse_pack_vector: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(nnz, 256), 256 * 8);
        PMT_LAUNCH(sparse_pack_vector_kernel, dim3(blocks), dim3(256), 0, s, nzval, perm, term_row, term_var, nnz, varmap, row_offset, out_terms);
        return check_launch("sparse_pack_vector_kernel");
    });
}
