// Sparse constraint matrix (BASELINE config 5): C*x (+|-) d with C in CSC (Julia SparseMatrixCSC layout,
// 1-based Int64 colptr/rowval).  The reference has no sparse code path: matvecmul! (src/functions.jl:775-798)
// walks every (row, col) of an AbstractMatrix, so a sparse C yields explicit-zero terms.  This node emits the
// terms of the STRUCTURAL non-zeros only, in the reference's row-major order (row, then ascending column) —
// i.e. the reference output with the structural zeros removed.  The row-major permutation of the CSC
// non-zeros is computed once on the host at plan time (the sparsity pattern is fixed across re-evaluations);
// each re-evaluation is one gather + coalesced 24-byte AoS write: 8 (nzval) + 8 (perm) + 8 (row) + 8 (var) read,
// 24 written per non-zero.
#include <new>
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

// ---- block form: the CSC -> row-major transposition goes through LDS ------------------------------------------------------------
// What the slab kernel still pays for (profiles/r03_sparse_blocks.txt): its gather nzval[perm[t]] is one L2 request per TERM
// (3.4 M per launch at config 5; the same kernel with a coalesced read instead runs in 27.8 instead of 34.2 us) and pulls 81 MB
// through the fabric for 54 MB of algorithmic reads — a slab of nzval (3.4 MB) plus the index streams do not fit an XCD's 4 MB L2.
// Here the matrix is cut into blocks of SB_RB rows x cw columns.  Within a block the coefficients of one column are a contiguous run
// of nzval (rows ascend within a CSC column), so a workgroup reads its block with coalesced loads — every line of nzval leaves HBM
// once — into LDS in CSC order; the terms of a row that fall into the column band are a contiguous run of the row-major output, and
// are written from LDS (16-byte chunks).  34.2 -> 30.1 us at config 5.  Static per pattern (pmt_sparse_blocks_build, host, once):
//   desc[rb * cols + c]  = { first CSC position of column c at or below row rb * SB_RB, count in the block | LDS slot of the first << 16 }
//   idx[t]               = LDS slot of term t's coefficient | column inside the band << 16         (4 bytes per term; the slab kernel: 8)
//   band_ptr[row * (ncb + 1) + cb] = index of the first term of `row` whose column lies in band cb
// and the variable word of a term comes from a per-COLUMN array staged in LDS (col_var, with the optimizer's index map applied when
// given), not from a per-term stream.  Per non-zero: 8 + 4 read, 24 written, plus 8 bytes per (row block, column).
constexpr int SB_RB = 128;         // rows per block
constexpr int SB_CAP = 7168;       // coefficients of a block held in LDS (56 KB; two workgroups per CU)
constexpr int SB_MAXCW = 1024;     // widest column band
constexpr int SB_NT = 512;
constexpr int SB_WROWS = SB_RB / 8;     // rows per wave

template <bool VAT_OUT>
__global__ __launch_bounds__(SB_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void sparse_block_kernel(
    const double *__restrict__ nzval, const uint2 *__restrict__ desc, const uint32_t *__restrict__ idx, const int64_t *__restrict__ band_ptr,
    const int64_t *__restrict__ col_var, int64_t rows, int64_t cols, int64_t nnz, int cw, int nrb, int ncb, const int64_t *__restrict__ varmap,
    int64_t row_offset, unsigned long long *__restrict__ out, const double *__restrict__ d, int sign, double *__restrict__ out_consts) {
    typedef unsigned long long u64;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    constexpr int W = VAT_OUT ? 3 : 2;                           // 8-byte words per term
    constexpr int R = SB_WROWS;
    __shared__ double s_val[SB_CAP];
    __shared__ u64 s_vartab[SB_MAXCW];
    // the column descriptors (load phase) and the per-wave images of the runs being written (write phase) share one buffer
    constexpr int IMG = 64 * 3 + 2;
    __shared__ __attribute__((aligned(16))) u64 s_img[8][IMG];
    static_assert(sizeof(u64) * 8 * IMG >= sizeof(uint2) * SB_MAXCW, "descriptor table fits the image buffer");
    uint2 *s_desc = reinterpret_cast<uint2 *>(&s_img[0][0]);
    __shared__ uint32_t s_t[8][2 * R];                           // term indices fit 32 bits (pmt_sparse_blocks_width)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> block: all row blocks of a column band on ONE XCD (workgroups go round-robin over the 8 XCDs), vertically adjacent
    // blocks share the lines their column runs straddle
    const int b = blockIdx.x;
    const int cb = (b / (8 * nrb)) * 8 + (b & 7), rb = (b >> 3) % nrb;
    if (cb >= ncb) return;
    // the constants of the node, 0.0 (+|-) d[row]: by the blocks of the first column band (one launch less: 6 us of config 5's 43)
    if (cb == 0 && out_consts && tid < SB_RB) {
        const int64_t row = (int64_t)rb * SB_RB + tid;
        if (row < rows) out_consts[row] = signed_const(d ? d[row] : 0.0, d ? sign : 0);
    }
    const int64_t c0 = (int64_t)cb * cw;
    const int ncol = (int)min((int64_t)cw, cols - c0);
    const int64_t wrow0 = (int64_t)rb * SB_RB + wave * R;        // first row of this wave
    // column descriptors of the block (coalesced, through LDS) and the row bounds of this wave
    const uint2 *drow = desc + (int64_t)rb * cols + c0;
#pragma unroll
    for (int k = 0; k < SB_MAXCW / SB_NT; ++k) {
        const int c = tid + k * SB_NT;
        if (c < ncol) s_desc[c] = drow[c];
    }
    if (lane < 2 * R) {
        const int64_t row = wrow0 + (lane >> 1);
        s_t[wave][lane] = row < rows ? (uint32_t)band_ptr[row * (ncb + 1) + cb + (lane & 1)] : 0u;
    }
    __syncthreads();
    auto bounds = [&](int i, int64_t &a, int64_t &e) {
        a = (int64_t)__builtin_amdgcn_readfirstlane((int)s_t[wave][2 * i]) & 0xffffffffll;
        e = (int64_t)__builtin_amdgcn_readfirstlane((int)s_t[wave][2 * i + 1]) & 0xffffffffll;
    };
    // ---- load phase: 8 lanes per column, 64 columns per pass, two consecutive coefficients per lane and load (runs are 8-byte aligned
    // only); all loads of the block are issued before the first LDS store
    typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
    const int sub2 = 2 * (tid & 7), grp = tid >> 3;
    constexpr int NP = SB_MAXCW / 64;
    uint32_t dl[NP];                                             // count | LDS slot << 16 of this lane's column in pass q
    f64x2u v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int c = grp + 64 * q;
        const uint2 cd = c < ncol ? s_desc[c] : make_uint2(0u, 0u);
        dl[q] = cd.y;
        const int len = (int)(cd.y & 0xffffu);
        const int64_t p = (int64_t)cd.x + sub2;
        v[q].x = 0.0; v[q].y = 0.0;
        if (sub2 < len) {
            if (p + 1 < nnz) v[q] = *reinterpret_cast<const f64x2u *>(nzval + p);     // the second one may belong to the next block: not stored
            else v[q].x = nzval[p];
        }
    }
    // static index words of the first 64 terms of each of the wave's rows: issued behind the coefficients, consumed in the write phase
    uint32_t id[R];
    int64_t longest = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int64_t a, e;
        bounds(i, a, e);
        longest = max(longest, e - a);
        id[i] = a + lane < e ? idx[a + lane] : 0u;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int len = (int)(dl[q] & 0xffffu), lb = (int)(dl[q] >> 16);
        if (sub2 < len) s_val[lb + sub2] = v[q].x;
        if (sub2 + 1 < len) s_val[lb + sub2 + 1] = v[q].y;
        if (len > 16) {                                          // more than 16 of 128 rows in one column: rare
            const int c = grp + 64 * q;
            const int64_t p0 = (int64_t)s_desc[c].x;
            for (int k = 16 + (tid & 7); k < len; k += 8) s_val[lb + k] = nzval[p0 + k];
        }
    }
    for (int c = tid; c < ncol; c += SB_NT) {
        const int64_t v = col_var[c0 + c];
        s_vartab[c] = (u64)(VAT_OUT ? map_var(varmap, v) : v);
    }
    __syncthreads();

    // ---- write phase: wave w owns rows wrow0 .. wrow0 + 15; 64 terms of each per round (a second round only for rows with more).
    // The write-out is bound by its INSTRUCTION count, not by memory (profiles/r03_sparse_blocks.txt: with the words of every 16-byte chunk
    // worked out per chunk — a division and three conditional reads per word — it took 28 us for 80 MB), so it is split into two plain copies:
    // lane l assembles term l of the run (row | coefficient | variable) in a per-wave LDS image of the run, laid out with the run's own
    // 16-byte phase; then the wave copies the image out linearly, 16 bytes per lane and instruction.  LDS operations of one wave execute in
    // order, so the image can be rewritten for the next row without a wait.
    u64 *img = s_img[wave];
    const u64 vrow_base = (u64)(row_offset + wrow0 + 1);
    for (int64_t base = 0; base < longest; base += 64) {
        if (base > 0) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                int64_t a, e;
                bounds(i, a, e);
                id[i] = a + base + lane < e ? idx[a + base + lane] : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            asm volatile("" ::: "memory");                       // one row at a time: keeps the bounds of later rows out of registers
            int64_t a, e;
            bounds(i, a, e);
            const int64_t ts = a + base;
            const int cnt = (int)max((int64_t)0, min((int64_t)64, e - ts));
            if (cnt == 0) continue;                              // wave-uniform
            u64 *seg = out + ts * W;
            const int nwords = cnt * W;
            const int lead = (int)((reinterpret_cast<uintptr_t>(seg) >> 3) & 1);
            // word q of the run lives at img[lead + q]: chunk c = words lead + 2c, lead + 2c + 1 is 16-byte aligned both here and in HBM
            if (lane < cnt) {
                u64 *t = img + lead + W * lane;
                const u64 coeff = (u64)__double_as_longlong(s_val[id[i] & 0xffffu]), var = s_vartab[id[i] >> 16];
                if (VAT_OUT) { t[0] = vrow_base + i; t[1] = coeff; t[2] = var; }
                else { t[0] = coeff; t[1] = var; }
            }
            __builtin_amdgcn_wave_barrier();
            if (lead && lane == 0) seg[0] = img[1];              // word 0 of an odd-aligned run
#pragma unroll
            for (int j = 0; j < (64 * W + 1 + 127) / 128; ++j) {
                if (lead + 128 * j >= nwords) continue;          // wave-uniform
                const int q0 = lead + 128 * j + 2 * lane;
                if (q0 + 1 < nwords) *reinterpret_cast<u64x2 *>(seg + q0) = *reinterpret_cast<const u64x2 *>(img + lead + q0);
                else if (q0 < nwords) seg[q0] = img[lead + q0];
            }
            __builtin_amdgcn_wave_barrier();
        }
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
                                         int64_t *rows_out, int64_t *cols_out, int64_t *row_ptr) try {
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
} catch (const std::bad_alloc &) {
    return pmt::fail(PMT_OUT_OF_MEMORY, "sparse_rowmajor_order: out of host memory");
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

extern "C" int pmt_sparse_slab_ptr(int64_t rows, int64_t cols, int nslab, const int64_t *row_ptr, const int64_t *term_col, int64_t *slab_ptr) try {
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
} catch (const std::bad_alloc &) {
    return pmt::fail(PMT_OUT_OF_MEMORY, "sparse_slab_ptr: out of host memory");
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

// ---- block form: host-side structure (once per pattern) and launchers
extern "C" int pmt_sparse_blocks_width(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int *out_cw) try {
    PMT_REQUIRE(m >= 0 && n >= 0, PMT_DIMENSION_MISMATCH, "sparse_blocks_width: negative dimension");
    PMT_REQUIRE(colptr && out_cw, PMT_INVALID_ARGUMENT, "sparse_blocks_width: null pointer");
    *out_cw = 0;
    if (m == 0 || n == 0) return PMT_OK;
    PMT_REQUIRE(colptr[0] == 1, PMT_INVALID_ARGUMENT, "sparse_blocks_width: colptr must be 1-based");
    const int64_t nnz = colptr[n] - 1;
    if (nnz == 0 || nnz >= ((int64_t)1 << 32)) return PMT_OK;           // nothing to do / positions do not fit the 32-bit descriptors
    PMT_REQUIRE(rowval, PMT_INVALID_ARGUMENT, "sparse_blocks_width: null pointer");
    const int64_t nrb = cdiv(m, SB_RB);
    // The block form pays when a row's part of a column band is a run worth writing (>= 16 terms on average: the hosts' gate).  For a LARGE
    // table that necessary condition is checked with the widest band BEFORE anything is sized by m * n: a hypersparse pattern (1e6 x 1e6 with
    // a few entries per column) must not allocate its (row block, strip) table — about 1 GB there — only to be turned down; it keeps the
    // slab form (cw = 0).  Small tables are simply built: the width is a property of the pattern, the gate is the caller's.
    if (nrb * cdiv(n, 32) > ((int64_t)1 << 22) && nnz < 16 * m * cdiv(n, SB_MAXCW)) {
        for (int64_t p = 0; p < nnz; ++p)
            if (rowval[p] < 1 || rowval[p] > m) return fail(PMT_DIMENSION_MISMATCH, "sparse_blocks_width: row index out of range");
        return PMT_OK;
    }
    // non-zeros per (row block, 32-column strip); rows must ascend within a column (they do in a SparseMatrixCSC)
    const int64_t nstrip = cdiv(n, 32);
    std::vector<int32_t> cnt;
    try { cnt.assign((size_t)(nrb * nstrip), 0); }
    catch (const std::bad_alloc &) { return fail(PMT_OUT_OF_MEMORY, "sparse_blocks_width: out of host memory for the block table"); }
    for (int64_t c = 0; c < n; ++c) {
        int64_t prev = 0;
        for (int64_t p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p) {
            const int64_t r = rowval[p];
            if (r < 1 || r > m) return fail(PMT_DIMENSION_MISMATCH, "sparse_blocks_width: row index out of range");
            if (r <= prev) return PMT_OK;                                 // unsorted / duplicate rows: the block form does not apply (cw = 0)
            prev = r;
            cnt[(size_t)(((r - 1) / SB_RB) * nstrip + c / 32)]++;
        }
    }
    for (int cw = SB_MAXCW; cw >= 32; cw >>= 1) {
        const int64_t per = cw / 32;
        int64_t worst = 0;
        for (int64_t rb = 0; rb < nrb; ++rb)
            for (int64_t s0 = 0; s0 < nstrip; s0 += per) {
                int64_t tot = 0;
                for (int64_t k = s0; k < std::min(nstrip, s0 + per); ++k) tot += cnt[(size_t)(rb * nstrip + k)];
                worst = std::max(worst, tot);
            }
        if (worst <= SB_CAP) { *out_cw = cw; return PMT_OK; }
    }
    return PMT_OK;
} catch (const std::bad_alloc &) {
    return pmt::fail(PMT_OUT_OF_MEMORY, "sparse_blocks_width: out of host memory");
}

extern "C" int pmt_sparse_blocks_build(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, const int64_t *perm,
                                       const int64_t *term_col, const int64_t *row_ptr, int cw, uint64_t *desc, uint32_t *idx, int64_t *band_ptr) try {
    PMT_REQUIRE(m > 0 && n > 0, PMT_DIMENSION_MISMATCH, "sparse_blocks_build: empty matrix");
    PMT_REQUIRE(cw >= 32 && cw <= SB_MAXCW && (cw & (cw - 1)) == 0, PMT_INVALID_ARGUMENT, "sparse_blocks_build: bad band width");
    PMT_REQUIRE(colptr && rowval && perm && term_col && row_ptr && desc && idx && band_ptr, PMT_INVALID_ARGUMENT, "sparse_blocks_build: null pointer");
    const int64_t nnz = colptr[n] - 1;
    PMT_REQUIRE(nnz < ((int64_t)1 << 32), PMT_DIMENSION_MISMATCH, "sparse_blocks_build: 2^32 or more non-zeros");
    const int64_t nrb = cdiv(m, SB_RB), ncb = cdiv(n, cw);
    std::vector<uint32_t> slot((size_t)nnz);
    std::vector<int32_t> fill((size_t)nrb);
    for (int64_t c = 0; c < n; ++c) {
        if (c % cw == 0) std::fill(fill.begin(), fill.end(), 0);
        int64_t p = colptr[c] - 1;
        const int64_t pend = colptr[c + 1] - 1;
        for (int64_t rb = 0; rb < nrb; ++rb) {
            const int64_t p0 = p;
            while (p < pend && rowval[p] <= (rb + 1) * SB_RB) {
                PMT_REQUIRE(rowval[p] > rb * SB_RB, PMT_INVALID_ARGUMENT, "sparse_blocks_build: rows must ascend within a column");
                slot[(size_t)p] = (uint32_t)(fill[(size_t)rb] + (p - p0));
                ++p;
            }
            const int64_t len = p - p0;
            PMT_REQUIRE(fill[(size_t)rb] + len <= SB_CAP, PMT_DIMENSION_MISMATCH, "sparse_blocks_build: block exceeds the LDS capacity (use pmt_sparse_blocks_width)");
            desc[rb * n + c] = (uint64_t)(uint32_t)p0 | ((uint64_t)((uint32_t)len | ((uint32_t)fill[(size_t)rb] << 16)) << 32);
            fill[(size_t)rb] += (int32_t)len;
        }
        PMT_REQUIRE(p == pend, PMT_INVALID_ARGUMENT, "sparse_blocks_build: row index out of range");
    }
    for (int64_t r = 0; r < m; ++r) {
        int64_t t = row_ptr[r];
        const int64_t tend = row_ptr[r + 1];
        for (int64_t cb = 0; cb <= ncb; ++cb) {
            while (t < tend && term_col[t] - 1 < cb * cw) ++t;
            band_ptr[r * (ncb + 1) + cb] = (cb == ncb) ? tend : t;
        }
        for (int64_t u = row_ptr[r]; u < tend; ++u) idx[u] = slot[(size_t)perm[u]] | ((uint32_t)((term_col[u] - 1) % cw) << 16);
    }
    return PMT_OK;
} catch (const std::bad_alloc &) {
    return pmt::fail(PMT_OUT_OF_MEMORY, "sparse_blocks_build: out of host memory");
}

static int launch_sparse_blocks(bool vat, const double *nzval, const uint64_t *desc, const uint32_t *idx, const int64_t *band_ptr,
                                const int64_t *col_var, int64_t rows, int64_t cols, int64_t nnz, int cw, const int64_t *varmap, int64_t row_offset, void *out,
                                const double *d, int sign, double *out_consts, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "sparse_pack_blocks: negative dimension");
    PMT_REQUIRE(nnz >= 0 && nnz < ((int64_t)1 << 32), PMT_DIMENSION_MISMATCH, "sparse_pack_blocks: bad number of non-zeros");
    PMT_REQUIRE(sign >= -1 && sign <= 1 && (sign == 0 || d || !out_consts), PMT_INVALID_ARGUMENT, "sparse_pack_blocks: sign must be -1, 0 or +1 and needs d");
    if (rows > 0 && out_consts && (cols == 0 || nnz == 0)) return d && sign ? pmt_consts_f64(d, rows, sign, out_consts, stream)
        : dispatch(stream, [=](hipStream_t s) { PMT_HIP_CHECK(hipMemsetAsync(out_consts, 0, rows * sizeof(double), s)); return PMT_OK; });
    if (rows == 0 || cols == 0 || nnz == 0) return PMT_OK;
    PMT_REQUIRE(cw >= 32 && cw <= SB_MAXCW && (cw & (cw - 1)) == 0, PMT_INVALID_ARGUMENT, "sparse_pack_blocks: bad band width");
    PMT_REQUIRE(nzval && desc && idx && band_ptr && col_var && out, PMT_INVALID_ARGUMENT, "sparse_pack_blocks: null pointer");
    const int64_t nrb = cdiv(rows, SB_RB), ncb = cdiv(cols, cw);
    PMT_REQUIRE(cdiv(ncb, 8) * 8 * nrb < ((int64_t)1 << 31), PMT_DIMENSION_MISMATCH, "sparse_pack_blocks: too many blocks");
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)(cdiv(ncb, 8) * 8 * nrb);
        const uint2 *dd = reinterpret_cast<const uint2 *>(desc);
        if (vat) PMT_LAUNCH_NAMED("sparse_block_kernel<VAT>", (sparse_block_kernel<true>), dim3(blocks), dim3(SB_NT), 0, s, nzval, dd, idx, band_ptr, col_var,
                                  rows, cols, nnz, cw, (int)nrb, (int)ncb, varmap, row_offset, reinterpret_cast<unsigned long long *>(out), d, sign, out_consts);
        else PMT_LAUNCH_NAMED("sparse_block_kernel<LT>", (sparse_block_kernel<false>), dim3(blocks), dim3(SB_NT), 0, s, nzval, dd, idx, band_ptr, col_var,
                              rows, cols, nnz, cw, (int)nrb, (int)ncb, varmap, row_offset, reinterpret_cast<unsigned long long *>(out), d, sign, out_consts);
        return check_launch("sparse_block_kernel");
    });
}

extern "C" int pmt_sparse_pack_vector_blocks_f64(const double *nzval, const uint64_t *desc, const uint32_t *idx, const int64_t *band_ptr,
                                                 const int64_t *col_var, int64_t rows, int64_t cols, int64_t nnz, int cw, const int64_t *varmap,
                                                 int64_t row_offset, const double *d, int sign, pmt_vector_affine_term *out_terms,
                                                 double *out_consts, void *stream) {
    return launch_sparse_blocks(true, nzval, desc, idx, band_ptr, col_var, rows, cols, nnz, cw, varmap, row_offset, out_terms, d, sign, out_consts, stream);
}

extern "C" int pmt_sparse_assemble_blocks_f64(const double *nzval, const uint64_t *desc, const uint32_t *idx, const int64_t *band_ptr,
                                              const int64_t *col_var, int64_t rows, int64_t cols, int64_t nnz, int cw, const double *d, int sign,
                                              pmt_linear_term *out_terms, double *out_consts, void *stream) {
    return launch_sparse_blocks(false, nzval, desc, idx, band_ptr, col_var, rows, cols, nnz, cw, (const int64_t *)nullptr, 0, out_terms, d, sign, out_consts,
                                stream);
}
