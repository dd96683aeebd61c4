// canonicalize! ordering ON THE DEVICE (src/functions.jl:269-272, 381-386, sort_and_combine! src/util.jl:9-26): the
// permutation that sorts a term list by canonical key and the boundaries of the runs of equal keys, computed from the term buffer where
// it lies in HBM — no copy of the indices to the host, no single-threaded std::stable_sort (round 1: seconds for 10^7-10^8 literal terms,
// paid again whenever the model structure changes).  Same conventions as the host version in canon.hip: keys var / (min, max); duplicates
// are added in their ORIGINAL order (a stable LSD radix sort: rocprim::radix_sort_pairs, shipped with ROCm); a run of length one keeps
// the original (row, col).  The only thing that crosses PCIe is the number of distinct terms (8 bytes): the host sizes the output from it.
// Setup-time code (like the host version, it allocates its scratch and synchronises); the per-solve work stays pmt_segment_sum_f64.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

namespace pmt {

typedef unsigned long long u64;

__global__ void canon_keys_kernel(const char *__restrict__ terms, int64_t n, int term_bytes, u64 *__restrict__ keys, int64_t *__restrict__ vals,
                                  int *__restrict__ too_large) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t *idx = reinterpret_cast<const int64_t *>(terms + i * term_bytes + 8);
    u64 key;
    if (term_bytes == 16) {
        key = (u64)idx[0];
    } else {
        const u64 a = (u64)min(idx[0], idx[1]), b = (u64)max(idx[0], idx[1]);
        if (b >> 32) *too_large = 1;                      // (the packed key needs both indices below 2^32; the caller falls back to the host sort)
        key = (a << 32) | (b & 0xffffffffull);
    }
    keys[i] = key;
    vals[i] = i;
}

__global__ void canon_heads_kernel(const u64 *__restrict__ keys, int64_t n, unsigned char *__restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void canon_init_terms_kernel(const char *__restrict__ terms, int term_bytes, const int64_t *__restrict__ perm,
                                        const int64_t *__restrict__ seg_ptr, int64_t nseg, char *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const int64_t first = perm[seg_ptr[s]];
    const int64_t *idx = reinterpret_cast<const int64_t *>(terms + first * term_bytes + 8);
    double *oc = reinterpret_cast<double *>(out + s * term_bytes);
    int64_t *oi = reinterpret_cast<int64_t *>(out + s * term_bytes + 8);
    oc[0] = 0.0;
    if (term_bytes == 16) {
        oi[0] = idx[0];
    } else if (seg_ptr[s + 1] - seg_ptr[s] == 1) {        // kept as it is (util.jl:18-19)
        oi[0] = idx[0]; oi[1] = idx[1];
    } else {                                              // combine canonicalises (functions.jl:186-191)
        oi[0] = min(idx[0], idx[1]); oi[1] = max(idx[0], idx[1]);
    }
}

}  // namespace pmt

using namespace pmt;

// perm: int64[n], seg_ptr: int64[n + 1] (device).  *nseg_host receives the number of runs.  PMT_INVALID_ARGUMENT with the message
// "...use the host ordering" if an index does not fit the packed 64-bit key.
extern "C" int pmt_canonical_order_device(const void *terms, int64_t n, int term_bytes, int64_t *perm, int64_t *seg_ptr, int64_t *nseg_host,
                                          void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "canonical_order_device: negative length");
    PMT_REQUIRE(term_bytes == 16 || term_bytes == 24, PMT_INVALID_ARGUMENT, "canonical_order_device: term_bytes must be 16 (LinearTerm) or 24 (QuadraticTerm)");
    PMT_REQUIRE(nseg_host && seg_ptr && (n == 0 || (terms && perm)), PMT_INVALID_ARGUMENT, "canonical_order_device: null pointer");
    PMT_REQUIRE(!is_recording_handle(stream), PMT_STATE_ERROR, "canonical_order_device: setup-time call, needs a HIP stream (not a recording handle)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) {
        *nseg_host = 0;
        PMT_HIP_CHECK(hipMemsetAsync(seg_ptr, 0, sizeof(int64_t), s));
        return PMT_OK;
    }
    const size_t un = (size_t)n;
    size_t sort_bytes = 0, sel_bytes = 0;
    u64 *keys = nullptr;
    PMT_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys, perm, perm, un, 0, 64, s));
    unsigned char *head = nullptr;
    size_t *count = nullptr;
    PMT_HIP_CHECK(rocprim::select(nullptr, sel_bytes, rocprim::counting_iterator<int64_t>(0), head, seg_ptr, count, un, s));
    // scratch: keys in/out, values in, head flags, flag + count, rocPRIM temporaries
    const size_t tmp_bytes = std::max(sort_bytes, sel_bytes);
    char *scratch = nullptr;
    const size_t total = 3 * un * 8 + ((un + 15) / 16) * 16 + 64 + tmp_bytes;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&scratch), total);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PMT_OUT_OF_MEMORY, std::string("canonical_order_device: hipMalloc: ") + hipGetErrorString(e)); }
    keys = reinterpret_cast<u64 *>(scratch);
    u64 *keys2 = keys + un;
    int64_t *vals = reinterpret_cast<int64_t *>(keys2 + un);
    head = reinterpret_cast<unsigned char *>(vals + un);
    char *small = reinterpret_cast<char *>(head) + ((un + 15) / 16) * 16;
    int *too_large = reinterpret_cast<int *>(small);
    count = reinterpret_cast<size_t *>(small + 16);
    void *tmp = small + 64;
    int rc = PMT_OK;
    auto finish = [&](int code) { (void)hipStreamSynchronize(s); (void)hipFree(scratch); return code; };
    if (hipMemsetAsync(small, 0, 64, s) != hipSuccess) return finish(fail(PMT_HIP_ERROR, "canonical_order_device: hipMemsetAsync"));
    const unsigned blocks = (unsigned)cdiv(n, 256);
    PMT_LAUNCH(canon_keys_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const char *>(terms), n, term_bytes, keys, vals, too_large);
    if ((rc = check_launch("canon_keys_kernel"))) return finish(rc);
    size_t b = tmp_bytes;
    if (rocprim::radix_sort_pairs(tmp, b, keys, keys2, vals, perm, un, 0, 64, s) != hipSuccess) return finish(fail(PMT_HIP_ERROR, "rocprim::radix_sort_pairs"));
    PMT_LAUNCH(canon_heads_kernel, dim3(blocks), dim3(256), 0, s, keys2, n, head);
    if ((rc = check_launch("canon_heads_kernel"))) return finish(rc);
    b = tmp_bytes;
    if (rocprim::select(tmp, b, rocprim::counting_iterator<int64_t>(0), head, seg_ptr, count, un, s) != hipSuccess) return finish(fail(PMT_HIP_ERROR, "rocprim::select"));
    int flag = 0;
    size_t nseg = 0;
    if (hipMemcpyAsync(&flag, too_large, sizeof flag, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&nseg, count, sizeof nseg, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return finish(fail(PMT_HIP_ERROR, "canonical_order_device: read-back"));
    if (flag) return finish(fail(PMT_INVALID_ARGUMENT, "canonical_order_device: an index does not fit the packed 64-bit key (>= 2^32); use the host ordering"));
    const int64_t end = n;
    if (hipMemcpyAsync(seg_ptr + nseg, &end, sizeof end, hipMemcpyHostToDevice, s) != hipSuccess) return finish(fail(PMT_HIP_ERROR, "canonical_order_device: seg_ptr end"));
    *nseg_host = (int64_t)nseg;
    return finish(PMT_OK);
}

// out_terms[s] = (0.0, indices of run s) for s < nseg — the static part of the canonical function, written once
extern "C" int pmt_canonical_init_terms(const void *terms, int term_bytes, const int64_t *perm, const int64_t *seg_ptr, int64_t nseg, void *out_terms,
                                        void *stream) {
    PMT_REQUIRE(nseg >= 0, PMT_DIMENSION_MISMATCH, "canonical_init_terms: negative count");
    PMT_REQUIRE(term_bytes == 16 || term_bytes == 24, PMT_INVALID_ARGUMENT, "canonical_init_terms: term_bytes must be 16 or 24");
    if (nseg == 0) return PMT_OK;
    PMT_REQUIRE(terms && perm && seg_ptr && out_terms, PMT_INVALID_ARGUMENT, "canonical_init_terms: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(canon_init_terms_kernel, dim3((unsigned)cdiv(nseg, 256)), dim3(256), 0, s, reinterpret_cast<const char *>(terms), term_bytes, perm,
                   seg_ptr, nseg, reinterpret_cast<char *>(out_terms));
        return check_launch("canon_init_terms_kernel");
    });
}
