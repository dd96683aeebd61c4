// prune_zero!(f; atol) on the device (src/functions.jl:294-297, 409-413): keep, in order, the terms with abs(coeff) > atol.
// Not on the reference's solve path and not a hot kernel, so the stable stream compaction is rocPRIM's
// (rocprim::select, shipped with ROCm) rather than a hand-written scan; the number of surviving terms is data dependent and lands in
// device memory — the caller synchronises before reading it and sizing anything from it.
#include <cstring>

#include <rocprim/device/device_select.hpp>

#include "common.h"

namespace pmt {

struct KeepLT {
    double atol;
    __device__ bool operator()(const LT &t) const { return fabs(t.coeff) > atol; }
};
struct KeepQT {
    double atol;
    __device__ bool operator()(const QT &t) const { return fabs(t.coeff) > atol; }
};

template <typename T, typename Pred>
static int prune(const T *in, int64_t n, Pred pred, T *out, int64_t *out_count, void *workspace, size_t workspace_bytes, hipStream_t s,
                 size_t *needed) {
    size_t bytes = 0;
    hipError_t e = rocprim::select(nullptr, bytes, in, out, reinterpret_cast<size_t *>(out_count), (size_t)n, pred, s);
    if (e != hipSuccess) return fail(PMT_HIP_ERROR, std::string("rocprim::select (size query): ") + hipGetErrorString(e));
    if (needed) { *needed = bytes; return PMT_OK; }
    if (workspace_bytes < bytes) return fail(PMT_INVALID_ARGUMENT, "prune_zero: workspace too small (pmt_prune_zero_workspace_bytes)");
    e = rocprim::select(workspace, bytes, in, out, reinterpret_cast<size_t *>(out_count), (size_t)n, pred, s);
    if (e != hipSuccess) return fail(PMT_HIP_ERROR, std::string("rocprim::select: ") + hipGetErrorString(e));
    return PMT_OK;
}

}  // namespace pmt

using namespace pmt;

extern "C" size_t pmt_prune_zero_workspace_bytes(int64_t n, int term_bytes) {
    size_t bytes = 0;
    if (n <= 0) return 16;
    if (term_bytes == 24) (void)prune<QT>(nullptr, n, KeepQT{0.0}, nullptr, nullptr, nullptr, 0, nullptr, &bytes);
    else (void)prune<LT>(nullptr, n, KeepLT{0.0}, nullptr, nullptr, nullptr, 0, nullptr, &bytes);
    return bytes < 16 ? 16 : bytes;
}

extern "C" int pmt_prune_zero_f64(const void *terms, int64_t n, int term_bytes, double atol, void *out_terms, int64_t *out_count, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "prune_zero: negative length");
    PMT_REQUIRE(term_bytes == 16 || term_bytes == 24, PMT_INVALID_ARGUMENT, "prune_zero: term_bytes must be 16 (LinearTerm) or 24 (QuadraticTerm)");
    PMT_REQUIRE(out_count, PMT_INVALID_ARGUMENT, "prune_zero: null out_count");
    if (n == 0) {
        return dispatch(stream, [=](hipStream_t s) {
            PMT_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int64_t), s));
            return PMT_OK;
        });
    }
    PMT_REQUIRE(terms && out_terms && workspace, PMT_INVALID_ARGUMENT, "prune_zero: null pointer");
    return dispatch(stream, [=](hipStream_t s) {
        if (term_bytes == 24)
            return prune<QT>(reinterpret_cast<const QT *>(terms), n, KeepQT{atol}, reinterpret_cast<QT *>(out_terms), out_count, workspace, workspace_bytes, s,
                             nullptr);
        return prune<LT>(reinterpret_cast<const LT *>(terms), n, KeepLT{atol}, reinterpret_cast<LT *>(out_terms), out_count, workspace, workspace_bytes, s,
                         nullptr);
    });
}
