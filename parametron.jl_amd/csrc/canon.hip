// Generic canonicalize! for arbitrary term lists (src/functions.jl:269-272, 381-386; sort_and_combine! src/util.jl:9-26).
//
// On this path the variable indices of every term are fixed once the model is built (only coefficients change between
// re-evaluations), so the sort is done ONCE: pmt_canonical_order_* (host, plan time) produce the permutation that sorts the
// terms by canonical key and the boundaries of the runs of equal keys.  Per re-evaluation canonicalize! is then a segmented
// sum of coefficients (pmt_segment_sum_f64): out[s] = c[perm[p0]] + c[perm[p0+1]] + ...; indices of the output terms are static
// and written by the host when the node is created.
//
// Conventions reproduced from the reference:
//   * keys: LinearTerm -> var index; QuadraticTerm -> (min(row,col), max(row,col))  (functions.jl:270, 383)
//   * a run of length one keeps the ORIGINAL (row, col) of its term — sort_and_combine! only canonicalises inside `combine`
//     (util.jl:18-23, functions.jl:186-191)
//   * the reference sorts with the unstable Base.Sort.QuickSort, so the order in which duplicates are added is an artefact of
//     that algorithm; here duplicates are added in their original order (stable).  Coefficients agree to rounding.
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.h"

namespace pmt {

// one wave per segment: lanes stride the run in original order, then a fixed butterfly — deterministic
__global__ __launch_bounds__(256) void segment_sum_kernel(const char *__restrict__ in, int64_t in_stride, const int64_t *__restrict__ perm,
                                                          const int64_t *__restrict__ seg_ptr, int64_t nseg, char *__restrict__ out,
                                                          int64_t out_stride) {
    const int64_t seg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = seg_ptr[seg], p1 = seg_ptr[seg + 1];
    double acc = 0.0;
    bool any = false;
    for (int64_t p = p0 + lane; p < p1; p += 64) {
        const double c = *reinterpret_cast<const double *>(in + perm[p] * in_stride);
        acc = any ? acc + c : c;
        any = true;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double other = __shfl_down(acc, off, 64);
        const int other_any = __shfl_down(any ? 1 : 0, off, 64);
        if (lane + off < 64 && other_any) {
            acc = any ? acc + other : other;      // lanes without elements contribute nothing (not even +0.0)
            any = true;
        }
    }
    if (lane == 0) *reinterpret_cast<double *>(out + seg * out_stride) = acc;
}

template <typename Key, typename KeyOf>
static int64_t order_by_key(int64_t n, KeyOf key_of, int64_t *perm, int64_t *seg_ptr) {
    std::iota(perm, perm + n, (int64_t)0);
    std::stable_sort(perm, perm + n, [&](int64_t a, int64_t b) { return key_of(a) < key_of(b); });
    int64_t nseg = 0;
    for (int64_t p = 0; p < n; ++p) {
        if (p == 0 || key_of(perm[p - 1]) < key_of(perm[p])) seg_ptr[nseg++] = p;
    }
    seg_ptr[nseg] = n;
    return nseg;
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_canonical_order_affine(int64_t n, const int64_t *vars, int64_t *perm, int64_t *seg_ptr, int64_t *out_vars, int64_t *nseg) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "canonical_order_affine: negative length");
    PMT_REQUIRE(nseg && seg_ptr && (n == 0 || (vars && perm && out_vars)), PMT_INVALID_ARGUMENT, "canonical_order_affine: null pointer");
    const int64_t s = order_by_key<int64_t>(n, [&](int64_t i) { return vars[i]; }, perm, seg_ptr);
    for (int64_t k = 0; k < s; ++k) out_vars[k] = vars[perm[seg_ptr[k]]];
    *nseg = s;
    return PMT_OK;
}

extern "C" int pmt_canonical_order_quadratic(int64_t n, const int64_t *rows, const int64_t *cols, int64_t *perm, int64_t *seg_ptr,
                                             int64_t *out_rows, int64_t *out_cols, int64_t *nseg) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "canonical_order_quadratic: negative length");
    PMT_REQUIRE(nseg && seg_ptr && (n == 0 || (rows && cols && perm && out_rows && out_cols)), PMT_INVALID_ARGUMENT,
                "canonical_order_quadratic: null pointer");
    auto key = [&](int64_t i) { return std::make_pair(std::min(rows[i], cols[i]), std::max(rows[i], cols[i])); };
    const int64_t s = order_by_key<std::pair<int64_t, int64_t>>(n, key, perm, seg_ptr);
    for (int64_t k = 0; k < s; ++k) {
        const int64_t first = perm[seg_ptr[k]];
        if (seg_ptr[k + 1] - seg_ptr[k] == 1) { out_rows[k] = rows[first]; out_cols[k] = cols[first]; }     // kept as is (util.jl:18-19)
        else { out_rows[k] = key(first).first; out_cols[k] = key(first).second; }                             // combine canonicalises
    }
    *nseg = s;
    return PMT_OK;
}

extern "C" int pmt_segment_sum_f64(const void *in_terms, int64_t in_stride_bytes, const int64_t *perm, const int64_t *seg_ptr, int64_t nseg,
                                   void *out_terms, int64_t out_stride_bytes, void *stream) {
    PMT_REQUIRE(nseg >= 0, PMT_DIMENSION_MISMATCH, "segment_sum: negative segment count");
    PMT_REQUIRE(in_stride_bytes >= 8 && out_stride_bytes >= 8 && (in_stride_bytes % 8) == 0 && (out_stride_bytes % 8) == 0, PMT_INVALID_ARGUMENT,
                "segment_sum: strides must be multiples of 8 bytes");
    if (nseg == 0) return PMT_OK;
    PMT_REQUIRE(in_terms && perm && seg_ptr && out_terms, PMT_INVALID_ARGUMENT, "segment_sum: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(segment_sum_kernel, dim3((unsigned)cdiv(nseg, 4)), dim3(256), 0, s, reinterpret_cast<const char *>(in_terms), in_stride_bytes,
                   perm, seg_ptr, nseg, reinterpret_cast<char *>(out_terms), out_stride_bytes);
        return check_launch("segment_sum_kernel");
    });
}
