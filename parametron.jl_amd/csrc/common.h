// Shared host-side plumbing for libparametron_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>

#include "parametron_hip.h"

namespace pmt {

using LT = pmt_linear_term;
using QT = pmt_quadratic_term;
using VAT = pmt_vector_affine_term;
static_assert(sizeof(LT) == 16 && sizeof(QT) == 24 && sizeof(VAT) == 24, "Julia isbits layouts");

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define PMT_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::pmt::fail(PMT_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define PMT_REQUIRE(cond, code, msg)                 \
    do {                                             \
        if (!(cond)) return ::pmt::fail(code, msg);  \
    } while (0)

// A launch either runs now on `stream` or, when `stream` is a plan's recording handle, is
// appended to that plan's tape (plan.cpp).
using Launch = std::function<int(hipStream_t)>;
int dispatch(void *stream, Launch launch);

// Per-kernel timing (the analogue of the reference's per-node `findallocs` report, src/debug.jl:4-23): when enabled
// through pmt_profile_enable(), every launch is bracketed by HIP events on its own stream.
struct ProfScope {
    ProfScope(const char *name, hipStream_t s);
    ~ProfScope();
    const char *name_; hipStream_t s_; hipEvent_t e0_ = nullptr, e1_ = nullptr;
};
#define PMT_LAUNCH_NAMED(name, kernel, grid, block, shmem, s, ...)          \
    do {                                                                    \
        ::pmt::ProfScope _prof(name, s);                                    \
        hipLaunchKernelGGL(kernel, grid, block, shmem, s, __VA_ARGS__);     \
    } while (0)
#define PMT_LAUNCH(kernel, grid, block, shmem, s, ...) PMT_LAUNCH_NAMED(#kernel, kernel, grid, block, shmem, s, __VA_ARGS__)

inline int check_launch(const char *name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PMT_HIP_ERROR, std::string(name) + ": " + hipGetErrorString(e));
    return PMT_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 0.0 (+|-) v : the constant a zero!'d AffineFunction ends up with after add!/subtract! of a Number
// (src/functions.jl:244,452,474).  sign == 0 -> 0.0.
__device__ __forceinline__ double signed_const(double v, int sign) {
    return sign > 0 ? 0.0 + v : (sign < 0 ? 0.0 - v : 0.0);
}
__device__ __forceinline__ int64_t map_var(const int64_t *__restrict__ varmap, int64_t var) {
    return varmap ? varmap[var - 1] : var;
}

}  // namespace pmt
