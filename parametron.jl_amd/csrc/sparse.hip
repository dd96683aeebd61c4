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

// ---- XCD-aware form of the same scatter --------------------------------------------------------------------------------------
// The gather nzval[perm[t]] of the kernels above walks the CSC value array at random (row-major order of CSC storage): every 8-byte
// read pulls a whole cache line through the fabric (~8x amplification; 2.5 TB/s algorithmic at config 5).  Here the columns are cut
// into `nslab` slabs and workgroup b works on slab b % nslab: workgroups are dispatched round-robin over the 8 XCDs, so with
// nslab = 8 each XCD only ever touches one eighth of nzval (3.4 MB at config 5 — inside its 4 MB L2) and every line it pulls is
// eventually used in full.  Within a row the terms of a slab are contiguous in the row-major output, so a wave owns a (row, slab)
// segment: it stages 64 coefficients and variable indices in LDS and writes the 24-byte terms as 16-byte chunks.
// slab_ptr[row * (nslab + 1) + s] = index of the first term of `row` whose column lies in slab s (host, once: pmt_sparse_slab_ptr).
// Round 2: the kernel was LATENCY bound, not bandwidth bound (46 us = 2.9 TB/s at config 5): a wave walked its four row segments one
// 64-term chunk at a time, and every chunk is a chain of two dependent memory round trips (perm[t], then nzval[perm[t]]) before its
// store.  Now the four rows of a wave advance together ("rounds"): four independent perm / variable-index loads are in flight, then four
// gathers, and the index loads of the NEXT round are issued before the current round is staged and written.
constexpr int SP_ROWS_PER_WAVE = 4;

// IDX: element type of perm / term_var — int64_t (the reference's Int64 indices) or uint32_t (same values, half the index stream, for
// patterns and variable indices below 2^32: pmt_sparse_*_slabs_u32_f64).  62-64 VGPRs: eight waves per SIMD, i.e. all eight workgroups
// that config 5 puts on a CU are resident at once.
template <bool VAT_OUT, typename IDX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void sparse_slab_kernel(
    const double *__restrict__ nzval, const IDX *__restrict__ perm, const IDX *__restrict__ term_var, const int64_t *__restrict__ slab_ptr,
    int64_t rows, int nslab, const int64_t *__restrict__ varmap, int64_t row_offset, unsigned long long *__restrict__ out) {
    typedef unsigned long long u64;
    constexpr int W = VAT_OUT ? 3 : 2;                           // 8-byte words per term
    constexpr int R = SP_ROWS_PER_WAVE;
    __shared__ u64 s_coeff[4][64];
    __shared__ u64 s_var[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: row bounds live in SGPRs
    const int slab = blockIdx.x % nslab;
    const int64_t rowbase = ((int64_t)(blockIdx.x / nslab) * 4 + wave) * R;
    int64_t t0[R], t1[R];
    int64_t longest = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t row = rowbase + i;
        t0[i] = t1[i] = 0;
        if (row < rows) { t0[i] = slab_ptr[row * (nslab + 1) + slab]; t1[i] = slab_ptr[row * (nslab + 1) + slab + 1]; }
        longest = max(longest, t1[i] - t0[i]);
    }
    // index loads of round 0
    int64_t pidx[R], var[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t t = t0[i] + lane;
        const bool ok = t < t1[i];
        pidx[i] = ok ? (int64_t)perm[t] : 0;
        var[i] = ok ? (int64_t)term_var[t] : 1;
    }
    for (int64_t base = 0; base < longest; base += 64) {
        double val[R];
#pragma unroll
        for (int i = 0; i < R; ++i) val[i] = (t0[i] + base + lane < t1[i]) ? nzval[pidx[i]] : 0.0;      // four independent gathers
        int64_t pn[R], vn[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {                                                                   // next round's indices, in flight meanwhile
            const int64_t t = t0[i] + base + 64 + lane;
            const bool ok = t < t1[i];
            pn[i] = ok ? (int64_t)perm[t] : 0;
            vn[i] = ok ? (int64_t)term_var[t] : 1;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int64_t ts = t0[i] + base;
            const int cnt = (int)max((int64_t)0, min((int64_t)64, t1[i] - ts));
            if (cnt == 0) continue;                                                                     // wave-uniform
            if (lane < cnt) {
                s_coeff[wave][lane] = (u64)__double_as_longlong(val[i]);
                s_var[wave][lane] = (u64)(VAT_OUT ? map_var(varmap, var[i]) : var[i]);
            }
            __builtin_amdgcn_wave_barrier();                      // one wave: LDS writes above are visible to its own reads below
            const u64 rowword = (u64)(row_offset + rowbase + i + 1);
            wave_write_words<W>(out + ts * W, cnt, lane, [&](int q) -> u64 {
                const int t = q / W, f = q - W * t;
                if (VAT_OUT) return f == 0 ? rowword : (f == 1 ? s_coeff[wave][t] : s_var[wave][t]);
                return f == 0 ? s_coeff[wave][t] : s_var[wave][t];
            });
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i = 0; i < R; ++i) { pidx[i] = pn[i]; var[i] = vn[i]; }
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

extern "C" int pmt_sparse_slab_ptr(int64_t rows, int64_t cols, int nslab, const int64_t *row_ptr, const int64_t *term_col, int64_t *slab_ptr) {
    PMT_REQUIRE(rows >= 0 && cols >= 0 && nslab >= 1, PMT_DIMENSION_MISMATCH, "sparse_slab_ptr: bad dimensions");
    PMT_REQUIRE(row_ptr && slab_ptr, PMT_INVALID_ARGUMENT, "sparse_slab_ptr: null pointer");
    for (int64_t r = 0; r < rows; ++r) {
        int64_t t = row_ptr[r];
        const int64_t tend = row_ptr[r + 1];
        for (int s = 0; s <= nslab; ++s) {
            const int64_t first_col = (int64_t)s * cols / nslab + 1;            // 1-based first column of slab s (slab nslab: past the end)
            while (t < tend && term_col[t] < first_col) ++t;
            slab_ptr[r * (nslab + 1) + s] = (s == nslab) ? tend : t;
        }
    }
    return PMT_OK;
}

template <typename IDX>
static int launch_sparse_slab(bool vat, const double *nzval, const IDX *perm, const IDX *term_var, const int64_t *slab_ptr, int64_t rows,
                              int nslab, const int64_t *varmap, int64_t row_offset, void *out, void *stream) {
    PMT_REQUIRE(rows >= 0 && nslab >= 1 && nslab <= 64, PMT_DIMENSION_MISMATCH, "sparse_pack_slabs: bad dimensions");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE(nzval && perm && term_var && slab_ptr && out, PMT_INVALID_ARGUMENT, "sparse_pack_slabs: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)(cdiv(rows, 4 * SP_ROWS_PER_WAVE) * nslab);
        constexpr bool narrow = sizeof(IDX) == 4;
        if (vat) PMT_LAUNCH_NAMED(narrow ? "sparse_slab_kernel<VAT,u32>" : "sparse_slab_kernel<VAT>", (sparse_slab_kernel<true, IDX>), dim3(blocks), dim3(256), 0, s,
                                  nzval, perm, term_var, slab_ptr, rows, nslab, varmap, row_offset, reinterpret_cast<unsigned long long *>(out));
        else PMT_LAUNCH_NAMED(narrow ? "sparse_slab_kernel<LT,u32>" : "sparse_slab_kernel<LT>", (sparse_slab_kernel<false, IDX>), dim3(blocks), dim3(256), 0, s,
                              nzval, perm, term_var, slab_ptr, rows, nslab, varmap, row_offset, reinterpret_cast<unsigned long long *>(out));
        return check_launch("sparse_slab_kernel");
    });
}

extern "C" int pmt_sparse_pack_vector_slabs_f64(const double *nzval, const int64_t *perm, const int64_t *term_var, const int64_t *slab_ptr,
                                                int64_t rows, int nslab, const int64_t *varmap, int64_t row_offset,
                                                pmt_vector_affine_term *out_terms, void *stream) {
    return launch_sparse_slab(true, nzval, perm, term_var, slab_ptr, rows, nslab, varmap, row_offset, out_terms, stream);
}

extern "C" int pmt_sparse_assemble_slabs_f64(const double *nzval, const int64_t *perm, const int64_t *term_var, const int64_t *slab_ptr,
                                             int64_t rows, int nslab, pmt_linear_term *out_terms, void *stream) {
    return launch_sparse_slab(false, nzval, perm, term_var, slab_ptr, rows, nslab, (const int64_t *)nullptr, 0, out_terms, stream);
}

extern "C" int pmt_sparse_pack_vector_slabs_u32_f64(const double *nzval, const uint32_t *perm, const uint32_t *term_var, const int64_t *slab_ptr,
                                                    int64_t rows, int nslab, const int64_t *varmap, int64_t row_offset,
                                                    pmt_vector_affine_term *out_terms, void *stream) {
    return launch_sparse_slab(true, nzval, perm, term_var, slab_ptr, rows, nslab, varmap, row_offset, out_terms, stream);
}

extern "C" int pmt_sparse_assemble_slabs_u32_f64(const double *nzval, const uint32_t *perm, const uint32_t *term_var, const int64_t *slab_ptr,
                                                 int64_t rows, int nslab, pmt_linear_term *out_terms, void *stream) {
    return launch_sparse_slab(false, nzval, perm, term_var, slab_ptr, rows, nslab, (const int64_t *)nullptr, 0, out_terms, stream);
}
